// launch_group_encode_u16.hip -- see launch_group_encode.inc: the instantiations for samples of 9 to 16 bits.  Compiled for gfx950 only.
#define JLS_LAUNCH_WIDE 1
#include "launch_group_encode.inc"
