#!/bin/bash
# Developer / CI check: in the device assembly of the tile kernels, the loop that saves a suspended tile's
# optimiser state (global_store_dword ... sc1) must be followed by `s_waitcnt vmcnt(0)` BEFORE the s_barrier that
# precedes the ring push -- every storing wave drains its own stores.  Exits non-zero when a kernel fails the check.
# usage: scripts/check_suspend_drain.sh            (compiles both precisions, default builds, to /tmp/dis)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p /tmp/dis
F="-O3 -std=c++17 --offload-arch=gfx950 -I$ROOT/include -I$ROOT/gpsat_amd/csrc --cuda-device-only -S"
[ -n "$SKIP_COMPILE" ] || /opt/rocm/bin/hipcc $F $ROOT/gpsat_amd/csrc/gpsat_kernels.hip -o /tmp/dis/chk32.s &
[ -n "$SKIP_COMPILE" ] || /opt/rocm/bin/hipcc $F $ROOT/gpsat_amd/csrc/gpsat_kernels_f64.hip -o /tmp/dis/chk64.s &
wait
python3 - <<'PY'
import re, sys
bad = 0
for f in ("/tmp/dis/chk32.s", "/tmp/dis/chk64.s"):
    txt = open(f).read()
    kernels = re.split(r"\n(?=_ZN5gpsat\S*gp_tile_kernel\S*:)", txt)
    n = 0
    for k in kernels[1:]:
        name = k.split(":", 1)[0]
        body = k.split(".end_amdhsa_kernel")[0].split("\n")
        # the state save: dword stores with sc1 (agent-scope atomic stores) -- find each store, then walk forward to the
        # first s_barrier; a vmcnt(0) wait must lie in between
        idx = [i for i, l in enumerate(body) if re.search(r"global_store_dword\s.*\bsc1\b", l) and "dwordx" not in l]
        if not idx:
            print("FAIL", name, "no sc1 state stores found"); bad += 1; continue
        ok = True
        for i in idx:
            j = i + 1
            seen = False
            while j < len(body) and "s_barrier" not in body[j]:
                if re.search(r"s_waitcnt\s+vmcnt\(0\)", body[j]) or re.search(r"s_waitcnt\s+.*vmcnt\(0\)", body[j]):
                    seen = True
                if body[j].strip().startswith("s_endpgm"):
                    break
                j += 1
            if j < len(body) and "s_barrier" in body[j] and not seen:
                ok = False
        n += 1
        if not ok:
            print("FAIL", name); bad += 1
    print(f, "kernels checked:", n)
sys.exit(1 if bad else 0)
PY
echo "suspend-drain check passed"
