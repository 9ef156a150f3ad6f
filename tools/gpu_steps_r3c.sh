out=$1
R=$GRAFT_REPO_ROOT
step() { # name, seconds, command... ; a step that times out ends the call
  local name=$1 secs=$2; shift 2
  local t0=$(date +%s)
  timeout -k 10 $secs "$@"; local rc=$?
  echo "$name rc=$rc ($(( $(date +%s) - t0 )) s)" | tee -a $R/$out/summary.txt
  if [ $rc -ge 124 ]; then echo "stopping after $name" | tee -a $R/$out/summary.txt; exit 1; fi
}
step "panel stamps" 200 bash -c "python tools/panel_stamps.py > $out/panel_stamps.jsonl 2> $out/panel_stamps.err"
