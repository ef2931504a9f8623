// launch_pixels_u8.hip -- see launch_pixels.inc: the instantiations for samples of up to 8 bits.  Compiled for gfx950 only.
#define JLS_LAUNCH_WIDE 0
#include "launch_pixels.inc"
