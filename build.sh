#!/bin/bash
# Build libneusky_hip.so (gfx950) in-tree.  Used by __graft_entry__.build().
set -e
cd "$(dirname "$0")"
SRCS="neusky_amd/csrc/api.cpp $(ls neusky_amd/csrc/*.hip)"
mkdir -p build
OBJS=""
pids=""
for f in $SRCS; do
  o="build/$(basename $f).o"
  OBJS="$OBJS $o"
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ include/neusky_hip.h -nt "$o" ] || [ neusky_amd/csrc/common.h -nt "$o" ] || [ neusky_amd/csrc/chain.h -nt "$o" ]; then
    EXTRA=""
    # attention.hip: scalar (SGPR) operands feed plain v_fmac; the SLP vectoriser's v_pk_fma_f32 needs them copied into VGPR pairs first
    case "$f" in *attention.hip) EXTRA="-fno-slp-vectorize";; esac
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $EXTRA -c "$f" -o "$o" &
    pids="$pids $!"
  fi
done
for p in $pids; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o neusky_amd/libneusky_hip.so $OBJS
echo "built neusky_amd/libneusky_hip.so"
# the waitcnt rule on the final ISA of every kernel (tools/isa_lint.py: no instruction touches the destination of a load that no
# s_waitcnt covers -- hidden loads are invisible to the compiler's own bookkeeping): a violation fails the build
lint_rc=0
lint_out=$(python3 tools/isa_lint.py $OBJS) || lint_rc=$?
echo "$lint_out" | tail -3
exit $lint_rc
