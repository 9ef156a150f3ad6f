out=$1
R=$GRAFT_REPO_ROOT
step() { # name, seconds, command... ; a step that times out ends the call
  local name=$1 secs=$2; shift 2
  local t0=$(date +%s)
  timeout -k 10 $secs "$@"; local rc=$?
  echo "$name rc=$rc ($(( $(date +%s) - t0 )) s)" | tee -a $R/$out/summary.txt
  if [ $rc -ge 124 ]; then echo "stopping after $name" | tee -a $R/$out/summary.txt; exit 1; fi
}
step "bench" 300 bash -c "python bench.py --no-cpu-baseline > $out/bench.json 2> $out/bench.err"
step "bench n4096" 300 bash -c "python bench.py --steps 100 --warmup 10 --no-cpu-baseline --points 4096 > $out/bench_b32n4096.json 2> $out/bench_b32n4096.err"
cd /tmp
step "prof c2" 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof -o c2 -- python3 $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline
step "prof b32n4096" 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof -o b32n4096 -- python3 $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --points 4096
cd $R
rm -f $out/prof/*kernel_trace*
