#!/bin/bash
# builds tools/mb/chunk_compile_mb.cpp and runs it on one 2^18-op piece of the benchmark circuit (recycled wire indices)
cd /root/repo
/opt/rocm/bin/hipcc -O3 -std=c++17 -x hip --offload-arch=gfx950 -Ireverie_amd/csrc -Iinclude $EXTRA reverie_amd/csrc/compile.cpp reverie_amd/csrc/compile_par.cpp tools/mb/chunk_compile_mb.cpp -o /tmp/ccmb 2>&1 | grep -E "error" 
python - <<'PY'
import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import circuits
prog, wit, wc, st = circuits.layered_gf2(layers=153, p_and=0.5, recycle=True)
n = 1 << 18
prog[5 * n:6 * n].tofile("/tmp/piece.bin")
open("/tmp/piece.wires", "w").write(str(wc[1]))
PY
/tmp/ccmb /tmp/piece.bin $(cat /tmp/piece.wires) ${K:-4} ${THREADS:-1 8 24 48 96}
