#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_alpnet_gpu.py tests/test_protosam_gpu.py -x -q -m gpu 2>&1 | tail -3
for g in 0 auto; do echo "PSAM_HIPGRAPH=$g"; PSAM_HIPGRAPH=$g python3 tools/per_slice_vitb_profile.py coarse 2>&1 | tail -1; PSAM_HIPGRAPH=$g python3 tools/per_slice_vitb_profile.py full 2>&1 | tail -1; done
