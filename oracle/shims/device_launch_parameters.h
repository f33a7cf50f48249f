#pragma once
// device-side trap used by the reference's `prefiltered` assertion
#define __trap __builtin_trap
