out=$1
R=$GRAFT_REPO_ROOT
step() { # name, seconds, command... ; a step that times out ends the call
  local name=$1 secs=$2; shift 2
  local t0=$(date +%s)
  timeout -k 10 $secs "$@"; local rc=$?
  echo "$name rc=$rc ($(( $(date +%s) - t0 )) s)" | tee -a $R/$out/summary.txt
  if [ $rc -ge 124 ]; then echo "stopping after $name" | tee -a $R/$out/summary.txt; exit 1; fi
}
step "all gpu tests" 800 bash -c "python -m pytest tests -q -m gpu > $out/tests_all.log 2>&1"
step "smoke" 200 bash -c "python -c 'import __graft_entry__ as g; g.smoke()' > $out/smoke.log 2>&1"
step "bench" 300 bash -c "python bench.py > $out/bench.json 2> $out/bench.err"
