#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_protosam_gpu.py tests/test_sam_gpu.py tests/test_reference_records_gpu.py -x -q -m gpu 2>&1 | tail -3
for g in 0 auto 0 auto; do echo -n "PSAM_HIPGRAPH=$g  "; PSAM_HIPGRAPH=$g python3 tools/per_slice_profile.py 1 16 auto 5 2>&1 | tail -1; done
PSAM_HIPGRAPH=0 python3 tools/per_slice_vitb_profile.py full 2>&1 | tail -1; python3 tools/per_slice_vitb_profile.py full 2>&1 | tail -1
