for st in 0 1 2 3 4; do for ld in 0 1; do
hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -DSLQ_SWEEP_ST=$st -DSLQ_SWEEP_LDW=$ld -o /tmp/mbr_${st}_$ld scripts/microbench_reorth.hip 2>/dev/null && echo "== store flavour $st, w load $ld" && timeout -k 10 120 /tmp/mbr_${st}_$ld brief | grep -v "^----"
done; done
