/* sdt_knobs.h -- the environment switches of the library and its hosts, in three classes (C and C++):
 *
 *   operational   read in every build:          SDT_TIMING (phase and allocator timings on stderr), SDT_COMM_TIMEOUT_S / SDT_DRAIN_TIMEOUT_S
 *                 sdt_env()                     (deadlines of a multi-rank context's waits), SDT_NO_ARENA (device memory straight from the
 *                                               driver), SDT_PARSE_THREADS (parser threads of the hosts), SDT_SLOW_EXIT (leave through exit():
 *                                               profilers that write their files from an exit handler)
 *   test hooks    honoured only when            shrink a limit or force a path so that SMALL inputs go through it: chunk sizes, pool sizes,
 *                 SDT_TEST_HOOKS=1 is set       receive buffers, workgroup counts, thresholds, the host forms of device phases.  tests/conftest.py
 *                 sdt_test_env()                sets SDT_TEST_HOOKS for the whole suite; a production run ignores every one of them.
 *   tuning        compiled out unless the       A/B measurement switches (batch sizes, launch sizes, table load, alternative kernels' geometry):
 *                 unit is built -DSDT_TUNING    tools/ab_build.sh builds such variants into gpurun_ab/; the shipped library does not look at them.
 *                 sdt_tuning_env()
 *
 * A production process therefore reads seven variables (the six operational ones and SDT_TEST_HOOKS). */
#ifndef SDT_KNOBS_H
#define SDT_KNOBS_H
#include <stdlib.h>

static inline const char *sdt_env(const char *name) { return getenv(name); }

static inline const char *sdt_test_env(const char *name)
{
	const char *on = getenv("SDT_TEST_HOOKS");
	return on && on[0] == '1' ? getenv(name) : NULL;
}

static inline const char *sdt_tuning_env(const char *name)
{
#ifdef SDT_TUNING
	return getenv(name);
#else
	(void)name;
	return NULL;
#endif
}

static inline int sdt_knob_int(const char *v, int dflt) { return v && *v ? atoi(v) : dflt; }

#endif
