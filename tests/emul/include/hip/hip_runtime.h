// test-only shim: the emulator build resolves <hip/hip_runtime.h> to the host emulator.
#pragma once
#include "../../hip_emul.h"
