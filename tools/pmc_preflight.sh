#!/bin/bash
# usage: tools/pmc_preflight.sh <counters...>        does the profiler accept this --pmc set in ONE pass?  (rc 0 yes)
#        tools/pmc_preflight.sh --limits [BLOCK...]  how many counters of a block fit one pass (default block: TA)
# The set is tried on a trivial HIP program — one torch fill kernel — never on bench.py: a rejected set aborts the profiled
# process inside its first HIP call ("error code 38 ... exceeds the capabilities of the hardware"), before anything of igdsp runs.
# Every try sits under a 60 s timeout; nothing loops on a failure.
cd /tmp && export TMPDIR=/tmp
try() {   # try <counters...> -> rc of the profiled trivial program
    local d; d=$(mktemp -d /tmp/pmc_pf.XXXXXX)
    timeout -k 5 60 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$d" -- python3 -c "import torch; torch.zeros(1 << 20, device='cuda').sum().item()" > "$d/log" 2>&1
    local rc=$?
    if [ $rc -ne 0 ]; then grep -m2 -E "error code|exceeds|Could not" "$d/log"; fi
    rm -rf "$d"
    return $rc
}
if [ "$1" = "--limits" ]; then
    shift
    list=$(timeout -k 5 60 rocprofv3 -L 2>/dev/null)
    for blk in ${@:-TA}; do
        # base counters of the block as the profiler lists them (derived ones carry an Expression and expand to base counters)
        names=$(echo "$list" | awk -v b="$blk" '/^Counter_Name/ {n=$3} /^Block/ {if ($3 == b) print n}' | sort -u | head -8)
        set -- $names
        echo "block $blk: $# base counters tried, in growing sets: $names"
        ok=0; acc=()
        for c in $names; do
            acc+=("$c")
            if try "${acc[@]}" > /tmp/pmc_pf_last 2>&1; then ok=${#acc[@]}; echo "  ${#acc[@]} counter(s) [${acc[*]}]: accepted"
            else echo "  ${#acc[@]} counter(s) [${acc[*]}]: REJECTED  $(cat /tmp/pmc_pf_last | head -1)"; break; fi
        done
        echo "block $blk: at most $ok counter(s) per pass (of the ones tried)"
    done
    exit 0
fi
echo "# pre-flight on a trivial HIP program: $*"
try "$@"
rc=$?
echo "# pre-flight rc=$rc"
exit $rc
