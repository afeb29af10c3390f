// kernels_expand_sd.hip — the expand launches that compute the 5 x 5 RMS of their band image in registers (k_expand_fast<.., SD>,
// kernels_pyramid.hip) as a translation unit of their own, built with -fno-slp-vectorize (build.py NO_SLP): the window of six band rows
// takes 172 registers that way and 211 with the register pairs the SLP vectoriser builds for packed multiplies and adds, which on gfx950
// cost what two plain ones cost (profiles/r04_valu_cost.txt). The device code is kernels_pyramid.hip's, compiled here a second time with
// only launch_expand_sd() behind it.
#define MUSICA_PYRAMID_SD_ONLY 1
#include "kernels_pyramid.hip"
