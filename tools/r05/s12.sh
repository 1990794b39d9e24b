#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_s12; mkdir -p $O
timeout 900 python3 -m pytest tests/test_sam_gpu.py tests/test_amg_gpu.py -x -q > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log
tail -3 $O/pytest.log
timeout 300 python3 tools/r05/decoder_bench.py 2>&1 | grep -v amdgpu | tee $O/decoder_bench.log
PSAM_T2I_ALL=0 PSAM_UPSCALE_MFMA=0 timeout 300 python3 tools/r05/decoder_bench.py 2>&1 | grep -v amdgpu | tee -a $O/decoder_bench.log
PSAM_T2I_ALL=1 PSAM_UPSCALE_MFMA=0 timeout 300 python3 tools/r05/decoder_bench.py 2>&1 | grep -v amdgpu | tee -a $O/decoder_bench.log
for p in neck,patch neck patch; do
  echo "== PSAM_SPLIT_PARTS=$p config 4 batched"
  PSAM_SPLIT_PARTS=$p timeout 600 python3 tools/parity_stats.py 4 2>&1 | grep -v amdgpu.ids | tee -a $O/parity_stats.log
done
timeout 900 python3 bench.py --no-cpu-baseline --no-other-configs > $O/bench.json 2> $O/bench.log
python3 - <<'PY'
import json
j=json.load(open("gpurun_out/r05_s12/bench.json"))
print(j["value"], j["ms_per_step"], j["roofline"]["frac"], j["stage_ms_per_step"], j["per_slice_forward"]["value"])
PY
