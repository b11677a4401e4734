python bench_grad.py 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print(j.get('method'), j.get('ms_per_eval', j.get('ms_per_step')))
"
python scripts/small_batch.py f64 2>/dev/null | grep -v amdgpu | grep "systems   1 \|systems   9 " | head -12
