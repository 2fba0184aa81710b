python tools/soak.py 2>&1 | tail -2
for i in 1 2 3; do python bench.py --no-cpu 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read());print('C3', round(d['ms_per_step'],3), round(d['roofline']['frac'],4), d['fit'])"; done
python -m pytest tests -x -q -m gpu -p no:cacheprovider 2>&1 | grep -E "passed|failed" | tail -1
