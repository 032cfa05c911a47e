cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_engine_gpu.py -m gpu -x -q -k "operand_copies or fuse_update_flag_on_the_bf16 or c5" 2>&1 | tail -3
timeout 600 python bench.py --config c5 --steps 600 --warmup 50 2>&1 | grep -v amdgpu.ids | python -c "
import sys,json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['ms_per_step'], d.get('multi_gpu_shard_path',{}).get('ms_per_step'))
        print(' '.join('%s %.1f' % (k.replace('layer','L').replace('k_',''),v['us_per_step']) for k,v in d['kernels'].items() if 'dr' in k or 'grads' in k or 'bwd' in k))
"
