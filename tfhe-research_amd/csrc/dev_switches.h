// dev_switches.h -- every switch that makes a build compute WRONG BITS, in one place.
//
// The timing probes below remove a piece of the kernel's work to see what it costs (profiles/r0N_kernel_ab.txt); a
// library built with one of them passes no parity test.  They exist only in a build that says so:
//   * they can be turned on only together with -DTFHE_DEV_BUILD (tools/dev_build.sh adds it when a probe is asked for);
//     without it, defining any of them is a compile error;
//   * tfhe-research_amd/build.py -- the only recipe of the shipped library -- refuses to pass TFHE_DEV_BUILD or a probe;
//   * tfhe_version() of a dev build says "[DEV BUILD ...]" and names the probes that are on, and
//     tests/test_gpu_c_abi.py / tests/test_abi_surface.py assert that the shipped library's version string has no such tag.
#pragma once

#if !defined(TFHE_DEV_BUILD)
#if defined(TFHE_PROBE_HOT_KEY) || defined(TFHE_PROBE_NO_EXCHANGE_READS) || defined(TFHE_PROBE_NO_TRANSPOSE) || \
    defined(TFHE_PROBE_NO_TEAM_SYNC)
#error "TFHE_PROBE_* switches compute wrong bits: they need -DTFHE_DEV_BUILD (tools/dev_build.sh), which tfhe_version() reports"
#endif
#endif

// every key chunk is read from the first 4 KiB of the key (vector L1): the kernel without its key stream
#ifndef TFHE_PROBE_HOT_KEY
#define TFHE_PROBE_HOT_KEY 0
#endif
// the digit spectra are not read back from LDS in the multiply-accumulate
#ifndef TFHE_PROBE_NO_EXCHANGE_READS
#define TFHE_PROBE_NO_EXCHANGE_READS 0
#endif
// the register windows of a transform are not exchanged: the kernel without the LDS round trips of its transposes
#ifndef TFHE_PROBE_NO_TRANSPOSE
#define TFHE_PROBE_NO_TRANSPOSE 0
#endif
// the team barriers compiled out
#ifndef TFHE_PROBE_NO_TEAM_SYNC
#define TFHE_PROBE_NO_TEAM_SYNC 0
#endif

#define TFHE_ANY_PROBE (TFHE_PROBE_HOT_KEY || TFHE_PROBE_NO_EXCHANGE_READS || TFHE_PROBE_NO_TRANSPOSE || TFHE_PROBE_NO_TEAM_SYNC)
