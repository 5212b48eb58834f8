for i in 1 2 3; do
  timeout -k 10 200 python bench.py --workload hpe --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('default', d['ms_per_step'])"
  ISB_FUSE_FRONT=1 timeout -k 10 200 python bench.py --workload hpe --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fuse_front', d['ms_per_step'])"
done
