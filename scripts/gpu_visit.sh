#!/bin/bash
# One GPU-box visit = a list of named steps (replaces the per-visit scripts gpu_r05_[a-r].sh of round 5).
#   gpurun --timeout 1500 -- 'bash scripts/gpu_visit.sh <visit-name> <step> [<step> ...]'
# Output goes to gpurun_out/<visit-name>/ (merged back by gpurun); every step is wrapped in its own `timeout`.
# Steps:
#   suite            pytest -m gpu (whole suite, per-test durations)
#   pytest:<expr>    pytest -m gpu -k <expr>  (whole tests/ directory)
#   file:<name>      pytest -m gpu tests/<name>
#   bench            python bench.py  (default K / W)
#   bench_driver     python bench.py --steps 20 --warmup 5  (the driver's call, timed)
#   prof             rocprofv3 --kernel-trace --stats of the default bench (no extras) -> <out>/prof
#   configs          scripts/config_bench.py (the five BASELINE configs)
#   elems            scripts/elem_bench.py (other element shapes / forms)
#   shards           scripts/shard_bench.py + exchange_overhead.py + large_shards.py
#   py:<script> ...  python scripts/<script> (arguments after ':' separated by ',')
#   sh:<command>     any shell command (quote it)
V=${1:?visit name}; shift
OUT=$PWD/gpurun_out/$V
mkdir -p $OUT
export TMPDIR=/tmp
T=${HPV_STEP_TIMEOUT:-1500}
for step in "$@"; do
  echo "=== [$V] $step"
  case "$step" in
    suite) timeout $T python -m pytest tests -m gpu -q --durations=60 > $OUT/pytest_gpu.log 2>&1; tail -8 $OUT/pytest_gpu.log ;;
    pytest:*) timeout $T python -m pytest tests -m gpu -q -x -k "${step#pytest:}" > $OUT/pytest_k.log 2>&1; tail -15 $OUT/pytest_k.log ;;
    file:*) n=${step#file:}; timeout $T python -m pytest tests/$n -m gpu -q --durations=15 > $OUT/pytest_${n%.py}.log 2>&1; tail -15 $OUT/pytest_${n%.py}.log ;;
    bench) timeout $T python bench.py 2> $OUT/bench.err | tail -1 > $OUT/bench.json; cut -c1-400 $OUT/bench.json ;;
    bench_driver) ( time timeout $T python bench.py --steps 20 --warmup 5 2> $OUT/bench_driver_style.err | tail -1 > $OUT/bench_driver_style.json ) 2> $OUT/bench_driver_style.time
                  cut -c1-300 $OUT/bench_driver_style.json; cat $OUT/bench_driver_style.time ;;
    prof) ( cd /tmp && timeout $T rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python $OLDPWD/bench.py --no-extras --no-cpu-baseline --no-residual-roofline --no-pmc > $OUT/prof_bench.json 2> $OUT/prof.err )
          f=$(find $OUT/prof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && head -8 "$f" ;;
    configs) timeout $T python scripts/config_bench.py > $OUT/config_bench.md 2> $OUT/config_bench.err; cat $OUT/config_bench.md ;;
    elems) timeout $T python scripts/elem_bench.py > $OUT/elem_bench.md 2> $OUT/elem_bench.err; cat $OUT/elem_bench.md ;;
    shards) for s in shard_bench exchange_overhead large_shards; do timeout $T python scripts/$s.py > $OUT/$s.md 2>&1; cat $OUT/$s.md; done ;;
    py:*) a=${step#py:}; IFS=, read -r -a av <<< "$a"; n=$(basename ${av[0]} .py); timeout $T python scripts/${av[0]} "${av[@]:1}" > $OUT/$n.log 2> $OUT/$n.err; tail -40 $OUT/$n.log ;;
    sh:*) timeout $T bash -c "${step#sh:}" > $OUT/sh_$(date +%s).log 2>&1; tail -40 $OUT/sh_*.log | tail -40 ;;
    *) echo "unknown step $step" ;;
  esac
done
