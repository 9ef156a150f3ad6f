out=$1
R=$GRAFT_REPO_ROOT
step() { # name, seconds, command... ; a step that times out ends the call
  local name=$1 secs=$2; shift 2
  timeout -k 10 $secs "$@"; local rc=$?
  echo "$name rc=$rc" | tee -a $R/$out/summary.txt
  if [ $rc -ge 124 ]; then echo "stopping after $name" | tee -a $R/$out/summary.txt; exit 1; fi
}
step "all gpu tests" 1100 bash -c "python -m pytest tests -q -m gpu > $out/tests_all.log 2>&1"
step "bench" 300 bash -c "python bench.py > $out/bench.json 2> $out/bench.err"
step "bench c3" 300 bash -c "python bench.py --points 2048 --profile final --no-cpu-baseline > $out/bench_c3.json 2> $out/bench_c3.err"
cd /tmp
step "prof" 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof -o c2 -- python3 $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline
step "prof c3" 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof -o c3 -- python3 $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --points 2048 --profile final
step "pmc fetch" 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/$out/pmc_fetch -o f -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-graph
step "pmc write" 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/$out/pmc_write -o w -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-graph
cd $R
step "pmc summary" 60 python tools/pmc_summary.py $out/pmc_fetch $out/pmc_write $out/pmc r2p
rm -rf $out/pmc_fetch/*kernel_trace* $out/pmc_write/*kernel_trace*
ls -la $out/pmc_fetch $out/pmc_write | head -20
