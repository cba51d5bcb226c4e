for a in 2 16 32 64 512; do echo "== align $a"; TTM_BAND_ROWALIGN=$a ONLY=all python tools/band_check.py 1000000 --no-oracle 2>&1 | grep "k_band\|vs k_"; done
