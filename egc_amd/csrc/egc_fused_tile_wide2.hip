// The WIDE one-launch batch kernel (egc_fused_tile_wide.inc) for layers with 2 k-slabs of 128 per chunk (128 < F_in <= 256).
#define EGC_FTW_NS 2
#include "egc_fused_tile_wide.inc"
