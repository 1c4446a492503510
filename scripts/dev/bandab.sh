export IPX_LIB=$PWD/image-processing_amd/lib_dev/libipx.so
for ln in 128 256 512; do echo lanes $ln; IPX_BAND_LANES=$ln ROWS=${ROWS_LIST} timeout -k 10 200 python scripts/dev/bandab.py || exit 1; done
