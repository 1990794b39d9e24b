cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests/test_sam_gpu.py tests/test_amg_gpu.py tests/test_protosam_gpu.py tests/test_reference_records_gpu.py -x -q 2>&1 | tail -3
timeout 300 python3 tools/r05/decoder_bench.py 2>&1 | grep -v amdgpu
timeout 900 python3 bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | python3 -c "
import json,sys; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['stage_ms_per_step'])"
