python -m pytest tests -m gpu -x -q -k "cg or pin or golden or chain or operator" 2>&1 | tail -8
for m in 0 1; do
  BBX_CG_MERGE_UPDATE=$m python bench.py --steps 50 --warmup 5 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('merge=$m', d['value'], d['ms_per_step'], d['config'].get('mean_n_cg_iter'), d['roofline']['operator'])"
done
for m in 0 1; do
  BBX_CG_MERGE_UPDATE=$m python bench.py --config config2 --steps 50 --warmup 5 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('config2 merge=$m', d['value'], d['ms_per_step'], d['config'].get('mean_n_cg_iter'))"
done
