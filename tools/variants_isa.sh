#!/bin/bash
# Static evidence for the opt-in kernel variants (no GPU): registers, spills, instruction mix of the hottest loop, MFMA spread
# and the in-order issue model (tools/isa_loop.py) -- default kernel vs variant.   tools/variants_isa.sh > profiles/r06_variants_isa.txt
set -e
R=$(cd "$(dirname "$0")/.." && pwd); W=$(mktemp -d)
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -I$R/include -I$R/speech2affective_gestures_amd/csrc"
asm() { /opt/rocm/bin/hipcc $FLAGS $R/speech2affective_gestures_amd/csrc/$1.hip -o $W/$1.s 2>/dev/null; }
echo "# csrc digest $(python3 $R/tools/csrc_digest.py) | $(/opt/rocm/bin/hipcc --version | grep -m1 -o 'HIP version.*')"
echo "# static only: no clock has seen these kernels (GPU access closed in r06); tools/ab_variants.py is the measurement"
asm wgrad_tr; asm wgrad_tr32p; asm tcn_fused32; asm tcn32p
echo; echo "== WGRAD32_PIPE: fp32-operand weight gradients, 160 x 160 tile, 2 bf16 pieces per operand (the fp32 step); 6 steps of 32 rows per loop iteration"
echo "-- default  csrc/wgrad_tr.hip wgrad_tr32_k<160,160,3,1,2,2,2>"
python3 $R/tools/isa_loop.py $W/wgrad_tr.s wgrad_tr32_kILi160ELi160ELi3ELi1ELi2ELi2ELi2E 6
echo "-- variant  csrc/wgrad_tr32p.hip wgrad_tr32p_k<2,3>   (WGRAD32_PIPE=1: three register sets)"
python3 $R/tools/isa_loop.py $W/wgrad_tr32p.s wgrad_tr32p_kILi2ELi3E 6
echo "-- variant  csrc/wgrad_tr32p.hip wgrad_tr32p_k<2,2>   (WGRAD32_PIPE=2: two register sets)"
python3 $R/tools/isa_loop.py $W/wgrad_tr32p.s wgrad_tr32p_kILi2ELi2E 6
echo; echo "== the same, 1 piece per operand (bf16 step mode)"
echo "-- default  wgrad_tr32_k<160,160,3,1,2,2,1>"
python3 $R/tools/isa_loop.py $W/wgrad_tr.s wgrad_tr32_kILi160ELi160ELi3ELi1ELi2ELi2ELi1E 6
echo "-- variant  wgrad_tr32p_k<1,3>"
python3 $R/tools/isa_loop.py $W/wgrad_tr32p.s wgrad_tr32p_kILi1ELi3E 6
echo "-- variant  wgrad_tr32p_k<1,2>"
python3 $R/tools/isa_loop.py $W/wgrad_tr32p.s wgrad_tr32p_kILi1ELi2E 6
echo; echo "== TCN32_PAIR: clip-resident text TCN of the fp32 step; hottest loop = ONE conv (forward: 20 K tiles) / the two data-gradient convs of a block (backward: 2 x 20 K tiles)"
echo "-- default  csrc/tcn_fused32.hip tcn32_fwd_k  (ONE clip per workgroup: divide per-loop numbers by 1 clip)"
python3 $R/tools/isa_loop.py $W/tcn_fused32.s tcn32_fwd_k 1
echo "-- variant  csrc/tcn32p.hip tcn32p_fwd_k<2>   (TCN32_PAIR=1, TWO clips per workgroup: divide by 2 clips)"
python3 $R/tools/isa_loop.py $W/tcn32p.s tcn32p_fwd_kILi2E 1
echo "-- variant  csrc/tcn32p.hip tcn32p_fwd_k<1>   (TCN32_PAIR=2, the planes kernel with ONE clip per workgroup)"
python3 $R/tools/isa_loop.py $W/tcn32p.s tcn32p_fwd_kILi1E 1
echo "-- default  tcn32_bwd_k"
python3 $R/tools/isa_loop.py $W/tcn_fused32.s tcn32_bwd_k 1
echo "-- variant  tcn32p_bwd_k<2>  (two clips)"
python3 $R/tools/isa_loop.py $W/tcn32p.s tcn32p_bwd_kILi2E 1
echo "-- variant  tcn32p_bwd_k<1>  (one clip)"
python3 $R/tools/isa_loop.py $W/tcn32p.s tcn32p_bwd_kILi1E 1
rm -rf $W
