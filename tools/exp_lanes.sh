for l in 2 3 4 2 3; do
  ISB_HPE_LANES=$l timeout -k 10 200 python bench.py --workload hpe --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lanes $l', d['ms_per_step'])"
done
for b in 384 512; do
  timeout -k 10 200 python bench.py --workload hpe --batch $b --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('batch $b', d['ms_per_step'], d['value'])"
  ISB_HPE_MICROBATCH=$b timeout -k 10 200 python bench.py --workload hpe --batch $b --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('batch $b microbatch $b', d['ms_per_step'], d['value'])"
done
