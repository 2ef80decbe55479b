#!/bin/bash
# tools/clock_watch.sh <logfile> -- <command...> : poll rocm-smi (clocks, power) every 0.25 s while <command> runs
log=$1; shift; shift
( while true; do echo "t=$(date +%s.%N)"; rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power|fclk"; sleep 0.25; done ) > "$log" 2>&1 &
watch=$!
"$@"
rc=$?
kill $watch
exit $rc
