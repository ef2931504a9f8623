// launch_pixels_u16.hip -- see launch_pixels.inc: the instantiations for samples of 9 to 16 bits.  Compiled for gfx950 only.
#define JLS_LAUNCH_WIDE 1
#include "launch_pixels.inc"
