out=$1
R=$GRAFT_REPO_ROOT
step() { # name, seconds, command... ; a step that times out ends the call
  local name=$1 secs=$2; shift 2
  local t0=$(date +%s)
  timeout -k 10 $secs "$@"; local rc=$?
  echo "$name rc=$rc ($(( $(date +%s) - t0 )) s)" | tee -a $R/$out/summary.txt
  if [ $rc -ge 124 ]; then echo "stopping after $name" | tee -a $R/$out/summary.txt; exit 1; fi
}
step "model + train tests" 800 bash -c "python -m pytest tests/test_gpu_model.py tests/test_gpu_train.py tests/test_gpu_parity_configs.py -q -m gpu > $out/tests.log 2>&1"
step "bench" 300 bash -c "python bench.py --no-cpu-baseline > $out/bench.json 2> $out/bench.err"
cd /tmp
step "prof c2" 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof -o c2 -- python3 $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline
cd $R
rm -f $out/prof/*kernel_trace*
