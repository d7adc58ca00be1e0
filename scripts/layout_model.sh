# Per-rank step of every (query groups x row shards) layout of an N-GPU strong-scaling run of the 1M gallery, 1024-query
# batches, measured on ONE GPU: a rank of layout Gq x Gs answers 1024/Gq queries against 1005994/Gs rows; with Gs > 1 the
# two-phase protocol runs with its collectives on a one-rank RCCL group (--force-protocol).  bash scripts/layout_model.sh
# Also writes gpurun_out/layout_model.json; copied to profiles/layout_model.json it is what bench.py --gpus N reads
# (`multi_gpu.strong_scaling_vs_model`).
rm -f /tmp/layout_model_rows.txt
for cfg in "1 1" "2 1" "1 2" "4 1" "2 2" "1 4" "8 1" "4 2" "2 4" "1 8"; do
  set -- $cfg; gq=$1; gs=$2
  rows=$(( (1005994 + gs - 1) / gs )); q=$(( 1024 / gq ))
  extra=""; [ $gs -gt 1 ] && extra="--force-protocol --option rescore_grid_x=$(( 96 / gs ))"
  python bench.py --scale-10m off --no-cpu-baseline --async-tail 0 --rows $rows --queries $q $extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
n=$gq*$gs
print('GPUs %d  layout %d query groups x %d row shards: rank step %.4f ms for %d queries x %d rows -> job %.0f q/s' % (n, $gq, $gs, d['ms_per_step'], $q, $rows, 1024/d['ms_per_step']*1e3))
open('/tmp/layout_model_rows.txt','a').write(json.dumps({'gpus': n, 'gq': $gq, 'gs': $gs, 'queries': $q, 'rows': $rows, 'ms': d['ms_per_step']}) + '\n')
"
done
python -c "
import json
rows=[json.loads(l) for l in open('/tmp/layout_model_rows.txt')]
json.dump({'source': 'scripts/layout_model.sh on one MI355X (${LAYOUT_MODEL_TAG:-untagged})', 'gallery_rows': 1005994, 'entries': rows}, open('gpurun_out/layout_model.json','w'), indent=1)
"
