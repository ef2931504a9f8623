// launch_group_encode_u8.hip -- see launch_group_encode.inc: the instantiations for samples of up to 8 bits.  Compiled for gfx950 only.
#define JLS_LAUNCH_WIDE 0
#include "launch_group_encode.inc"
