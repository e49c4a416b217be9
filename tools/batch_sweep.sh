# images/s of the exact-f32 generator leg by batch size: per-call F(4x4) rule (default) / F(4x4) forced / F(2x2) only
export PYTHONPATH=$PWD
run() { python bench.py --only-headline --no-cpu-baseline --steps 20 "$@" 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$TAG', r['value'], r['ms_per_step'])"; }
for B in 1 2 4 8; do
  TAG="B=$B auto  "; run --batch $B
  TAG="B=$B forced"; run --batch $B --opt sean.wino4_force=1
  TAG="B=$B f2x2  "; run --batch $B --wino 1
done
