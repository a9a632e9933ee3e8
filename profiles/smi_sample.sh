# samples rocm-smi while a command runs: GPU use %, memory activity %, clocks, power
out=$1; shift
( while true; do /opt/rocm/bin/rocm-smi --showuse --showmemuse --showclocks --showpower --json 2>/dev/null | tr -d '\n'; echo; sleep 0.25; done ) > $out &
SMI=$!
"$@"
kill $SMI
