export IPX_LIB=$PWD/image-processing_amd/lib_dev/libipx.so
ROWS=${ROWS_LIST} timeout -k 10 200 python scripts/dev/bandab.py
