// dsp_shapes.hpp — compile-time facts about kernel shapes shared by the host preparation (dsp_prepare.hpp, plain C++) and the
// kernels (dsp_device.hpp).
#pragma once

namespace dsp {

// Shapes whose register-resident kernel takes ONE store address per exchange buffer + compile-time offsets: the host then
// builds the same slot map for every 64-position block (optimise_slots(shared)).  cpl / rpl = owned columns / rows per lane.
constexpr bool shared_slot_maps(int cpl, int rpl) { return cpl + rpl > 8; }   // (>= 6 measured: -1.5 % on the 24-h metric kernel, more gather conflicts)

// Register-resident solve kernels keep the per-scenario values that only their rare blocks touch (row bounds, norms, weight
// guards, ...: struct Rare in dsp_kernels.hip) in the wave's LDS region: 2 rpl + 9 lane-private doubles.
constexpr int rare_lds_bytes(int rpl) { return (2 * rpl + 9) * 512; }
// Every solve kernel stages the scale factors of the owned columns / rows behind the waves' buffers, once per block:
// [cpl + rpl][64] doubles (the KKT test measures its residuals in the unscaled space).
constexpr int scale_lds_bytes(int cpl, int rpl) { return (cpl + rpl) * 512; }

}  // namespace dsp
