export PYTHONPATH=$PWD
run() { python bench.py --only-headline --no-cpu-baseline --steps 20 "$@" 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$TAG', r['value'], r['ms_per_step'])"; }
for T in 64 128 256 512; do
  TAG="blocky w4ace=$T"; run --opt sean.wino4_ace=$T
  TAG="face   w4ace=$T"; run --opt sean.wino4_ace=$T --labels face
  TAG="dense  w4ace=$T"; run --opt sean.wino4_ace=$T --sparse 0
  TAG="pipe   w4ace=$T"; run --opt sean.wino4_ace=$T --workload pipeline --path f32
done
