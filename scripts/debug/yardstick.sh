# The measurements behind DESIGN 4.8 "the vendor's DGEMM as a yardstick" in one gpurun call (binaries: see the headers of the .hip files):
#   gpurun -- bash scripts/debug/yardstick.sh > profiles/rN_vendor_dgemm_and_tile_probes.txt
echo "== rocBLAS / Tensile DGEMM through torch.addmm (fp64), the solver's shapes  [scripts/debug/rocblas_dgemm_probe.py]"
timeout 300 python scripts/debug/rocblas_dgemm_probe.py 2>&1 | grep "^M="
echo; echo "== the K loop piece by piece, cache-resident data: the 256 x 128 eight-wave tile of esl_chol.hpp  [scripts/debug/tile_probe.hip]"
scripts/debug/bin/tile_probe
echo; echo "== the same for the 128 x 128 four-wave tile (two workgroups per CU / two halves of one workgroup)  [scripts/debug/tile_probe_v.hip]"
scripts/debug/bin/tile_probe_v
echo; echo "== C -= X^T X, n = 18,000, K = 3,744: k_chol_update_lds<256,128> against k_chol_update_v on the same data, bit for bit  [scripts/debug/upd_v_probe.hip]"
echo "-- dense X"; scripts/debug/bin/upd_v_probe 18000 3744 0 0
echo "-- X with a staircase of leading zero rows (12 % of the products skipped, as esl_cf.hpp's)"; scripts/debug/bin/upd_v_probe 18000 3744 1 0
echo "-- tile order by 8 x 8 blocks per XCD"; scripts/debug/bin/upd_v_probe 18000 3744 0 8 | grep "^v("
echo "-- n = 16,384, K = 512 (a trailing update of the factorisation)"; scripts/debug/bin/upd_v_probe 16384 512 0 0
echo; echo "== PMC passes over the dense-X run (per dispatch, medians)  [scripts/debug/pmc_probe.sh]"
bash scripts/debug/pmc_probe.sh gpurun_out/pmc_y scripts/debug/bin/upd_v_probe 18000 3744 0 0 > /dev/null 2>&1
grep "k_chol_update" gpurun_out/pmc_y/summary.txt
