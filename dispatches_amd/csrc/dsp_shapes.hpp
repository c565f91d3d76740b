// dsp_shapes.hpp — compile-time facts about kernel shapes shared by the host preparation (dsp_prepare.hpp, plain C++) and the
// kernels (dsp_device.hpp).
#pragma once

namespace dsp {

// Shapes whose register-resident kernel takes ONE store address per exchange buffer + compile-time offsets: the host then
// builds the same slot map for every 64-position block (optimise_slots(shared)).  cpl / rpl = owned columns / rows per lane.
constexpr bool shared_slot_maps(int cpl, int rpl) { return cpl + rpl > 8; }   // (>= 6 measured: -1.5 % on the 24-h metric kernel, more gather conflicts)

}  // namespace dsp
