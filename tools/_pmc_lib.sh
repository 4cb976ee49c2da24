# tools/_pmc_lib.sh — sourced by the profiling scripts: one rocprofv3 pass that cannot leave a stale or partial result
# behind unnoticed. The output directory is removed first; the pass's exit code and the presence of its CSV
# are checked; a failure prints a line and sets PROF_RC=1 (the calling script exits with it).
#   pmc_pass   DIR COUNTER... -- PROGRAM ARGS...     counters only (never together with a trace domain)
#   trace_pass DIR -- PROGRAM ARGS...                --kernel-trace --stats
# The program behind `--` is python3 itself (no env / bash -c hop: the profiler's preload has initialised the GPU).
PROF_RC=0
pmc_pass() {
  local d=$1; shift
  local ctr=()
  while [ "$1" != "--" ]; do ctr+=("$1"); shift; done
  shift
  rm -rf "$d"; mkdir -p "$d"
  rocprofv3 --pmc "${ctr[@]}" --output-format csv -d "$d" -o c -- "$@" > "$d.log" 2>&1
  local e=$?
  if [ $e -ne 0 ] || [ -z "$(find "$d" -name '*counter_collection.csv' 2>/dev/null | head -1)" ]; then
    echo "PROFILE PASS FAILED: $d (rc $e, counters ${ctr[*]}; log $d.log)"; PROF_RC=1
  fi
}
trace_pass() {
  local d=$1; shift; shift
  rm -rf "$d"; mkdir -p "$d"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$d" -o kt -- "$@" > "$d.out" 2> "$d.log"
  local e=$?
  if [ $e -ne 0 ] || [ -z "$(find "$d" -name '*kernel_stats.csv' 2>/dev/null | head -1)" ]; then
    echo "PROFILE PASS FAILED: $d (rc $e; log $d.log)"; PROF_RC=1
  fi
}
