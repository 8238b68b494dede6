// Configuration.h -- scalar types and error macro of the host classes (reference: src/Configuration.h:30-81).
// real_t is double here: the reference is fp32-only (Configuration.h:31); the device precision is chosen at run
// time (RN_F32 / RN_F64), the host API is always double.
#ifndef RAPIDNET_HOST_CONFIGURATION_H_
#define RAPIDNET_HOST_CONFIGURATION_H_

#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <stdexcept>
#include <string>

typedef int uint_t;   // sic: the reference's uint_t is int (Configuration.h:30)
typedef double real_t;

using std::string;

// The reference prints and exit()s on a failed check (Configuration.h:70-81).  A library must not kill its host
// process: the same condition throws, and the top-level drivers decide.
#define _ASSERT(cond)                                                                                        \
    do {                                                                                                     \
        if (!(cond)) {                                                                                       \
            throw std::logic_error(std::string("assertion failed: ") + #cond + " at " + __FILE__ + ":" + std::to_string(__LINE__)); \
        }                                                                                                    \
    } while (0)

#endif
