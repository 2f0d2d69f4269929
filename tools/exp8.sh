export TMPDIR=/tmp
for kz in 8 15 22 33; do TP_TILE_KZ=$kz timeout 120 python tools/fine_ab.py 128 128 128 2>&1 | tail -n 1; done
for kz in 16 32 65 129; do TP_TILE_KZ=$kz timeout 120 python tools/fine_ab.py 256 256 256 2>&1 | tail -n 1; done
for kz in 4 5 6 8 11 13; do TP_TILE_KZ=$kz timeout 120 python tools/fine_ab.py 128 64 64 2>&1 | tail -n 1; done
for kz in 4 6 8 11 16; do TP_TILE_KZ=$kz timeout 120 python tools/fine_ab.py 192 64 64 2>&1 | tail -n 1; done
for kz in 8 16 26 43 65; do TP_TILE_KZ=$kz timeout 120 python tools/fine_ab.py 256 128 128 2>&1 | tail -n 1; done
