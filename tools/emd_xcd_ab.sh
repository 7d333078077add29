for x in 0 1; do echo "GENPC_EMD_XCD=$x"; GENPC_EMD_XCD=$x python tools/time_emd_grid.py 2>&1 | grep -E "uniform 13x16384|uniform 8x32768|13 scans|waymo|64x2048"; done
