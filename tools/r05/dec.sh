cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests/test_kernels_core_gpu.py tests/test_sam_gpu.py -x -q -k "layernorm or folded or image_encoder" 2>&1 | tail -3
timeout 900 python3 bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | python3 -c "
import json,sys; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['roofline']['frac'], j['stage_ms_per_step'], j['overlap_streams_auto']['value'], j['overlap_streams_auto']['single_stream_same_minute'])"
