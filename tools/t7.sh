python -m pytest tests/test_clip_gpu.py -q -x -k "tile_256x192" 2>&1 | tail -5
python tools/gemm_sweep.py 5,7,1 2>&1 | grep -E "9600x 2304|2400x 2304|512x 1536|8192"
for i in 1 2 3; do
echo "auto $(python bench.py --no-cpu-baseline --no-extras --steps 40 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")"
echo "E5_B=5 $(CC_TILE_E5_B=5 python bench.py --no-cpu-baseline --no-extras --steps 40 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")"
done
