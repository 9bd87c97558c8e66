// The WIDE one-launch batch kernel (egc_fused_tile_wide.inc) for layers with 3 k-slabs of 128 per chunk (256 < F_in <= 320).
#define EGC_FTW_NS 3
#include "egc_fused_tile_wide.inc"
