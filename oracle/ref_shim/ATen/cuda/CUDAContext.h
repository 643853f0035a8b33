// oracle shim header (test infrastructure): see cpu_cuda_shim.h
#pragma once
#include "../../cpu_cuda_shim.h"
