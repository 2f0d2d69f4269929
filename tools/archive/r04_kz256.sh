for kz in 0 16 24 32 64 86 129 257; do echo "TP_TILE_KZ=$kz"; TP_TILE_KZ=$kz python tools/cheb_variants.py 256 2>&1 | grep copy-only; done
echo "shape 16x16"; TP_FINE_SHAPE=1 python tools/cheb_variants.py 256 2>&1 | grep copy-only
echo "no xcd remap"; TP_XCD_REMAP=0 python tools/cheb_variants.py 256 2>&1 | grep copy-only
