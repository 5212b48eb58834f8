for cfg in "256 2" "128 1" "128 2" "256 1" "64 1" "256 2" "128 1"; do set -- $cfg
  ISB_HPE_MICROBATCH=$1 ISB_HPE_LANES=$2 timeout -k 10 200 python bench.py --workload hpe --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('microbatch $1 lanes $2', d['ms_per_step'], d['value'])"
done
