# developer helper (round 4): rawnet2 tests + per-layer profile of the f32x3 path
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_rawnet2.py tests/test_gpu_fullsize.py -x -q -m gpu > gpurun_out/r4_x3rn_tests.log 2>&1
tail -3 gpurun_out/r4_x3rn_tests.log
timeout -k 10 400 python tools/rn_layers.py rawnet2 f32x3 2>&1 | grep -v amdgpu.ids | grep -E "rn_step|split_s32|gemm_conv M27|gemm_conv M90|rn_sinc|rn_bn_act|total"
python bench.py --model rawnet2 --compute f32x3 --no-cpu-baseline --no-scoring --no-extras --steps 6 --warmup 2 --sustain-seconds 0 > gpurun_out/x3s.json 2> gpurun_out/x3s.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/x3s.json").read().strip().splitlines()[-1])
print("rawnet2 f32x3", round(d["value"]), round(d["ms_per_step"], 3), d["check"])
PY
