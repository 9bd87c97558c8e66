// The WIDE one-launch batch kernel (egc_fused_tile_wide.inc) for layers with 1 k-slab of 128 per chunk (0 < F_in <= 128).
#define EGC_FTW_NS 1
#include "egc_fused_tile_wide.inc"
