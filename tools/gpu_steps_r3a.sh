out=$1
R=$GRAFT_REPO_ROOT
step() { # name, seconds, command... ; a step that times out ends the call
  local name=$1 secs=$2; shift 2
  local t0=$(date +%s)
  timeout -k 10 $secs "$@"; local rc=$?
  echo "$name rc=$rc ($(( $(date +%s) - t0 )) s)" | tee -a $R/$out/summary.txt
  if [ $rc -ge 124 ]; then echo "stopping after $name" | tee -a $R/$out/summary.txt; exit 1; fi
}
step "all gpu tests" 900 bash -c "python -m pytest tests -q -m gpu > $out/tests_all.log 2>&1"
step "smoke" 200 bash -c "python -c 'import __graft_entry__ as g; g.smoke()' > $out/smoke.log 2>&1"
step "bench" 300 bash -c "python bench.py > $out/bench.json 2> $out/bench.err"
cd /tmp
step "prof c2" 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof -o c2 -- python3 $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline
step "pmc fetch" 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/$out/pmc_fetch -o f -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-graph
step "pmc write" 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/$out/pmc_write -o w -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-graph
cd $R
step "pmc summary" 60 python tools/pmc_summary.py $out/pmc_fetch $out/pmc_write $out/pmc r2
rm -f $out/pmc_fetch/*kernel_trace* $out/pmc_write/*kernel_trace* $out/prof/*kernel_trace*
