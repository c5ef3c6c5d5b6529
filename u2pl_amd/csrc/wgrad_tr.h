// csrc/wgrad_tr.hip: split-fp32 weight gradient with hardware-transposed LDS operand reads (see there).
#pragma once
#include "conv_geom.h"

bool wgrad_tr_eligible(const ConvGeom& g);
// slabs of the pixel range: ctiles = Cin tiles, nsplit slabs of cps 32-pixel chunks each; taps = R * S (or the batch)
void wgrad_tr_plan(const ConvGeom& g, int taps, int& ctiles, int& nsplit, int& cps);
// dy_amax / x_amax: NULL (six bf16 piece products) or the device-scalar maxima of the two operands (split-fp16, three products)
int launch_wgrad_tr(const float* dy, long lddy, const float* x, long ldx, float* part, const ConvGeom& g, int ctiles, int nsplit,
                    int cps, hipStream_t stream, long zdy, long zx, const float* dy_amax = nullptr, const float* x_amax = nullptr);
