/* Stand-in for <hip/hip_runtime.h> when the device sources are compiled for the CPU wave emulator
 * (tests/wave_emu, test infrastructure only: never part of the product library). */
#pragma once
#include "../wave_emu.h"
