/* CPUs this process may really use: min(online, affinity mask, cgroup CPU quota, OMP_NUM_THREADS).  A GPU box shows every core
 * of the host but gives a job a share; OpenMP teams of the host layer are sized to that share (a 128-thread team inside a
 * 16-CPU quota spends its time being throttled). */
#ifndef GLC_CPUS_H
#define GLC_CPUS_H
#ifndef _GNU_SOURCE
#define _GNU_SOURCE
#endif
#include <omp.h>
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>

static inline int glc_host_cpus_hw(void) {          /* affinity and cgroup quota: fixed for the life of the process */
    static int cached = 0;
    if (cached > 0) return cached;
    int n = omp_get_num_procs();
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) { int c = CPU_COUNT(&set); if (c > 0 && c < n) n = c; }
    FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r");                 /* cgroup v2: "<quota|max> <period>" */
    if (f) {
        char q[32]; long period = 0;
        if (fscanf(f, "%31s %ld", q, &period) == 2 && q[0] != 'm' && period > 0) {
            long quota = atol(q), c = (quota + period - 1) / period;
            if (c > 0 && c < n) n = (int)c;
        }
        fclose(f);
    } else if ((f = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r"))) {   /* cgroup v1 */
        long quota = -1, period = 0;
        if (fscanf(f, "%ld", &quota) != 1) quota = -1;
        fclose(f);
        FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r");
        if (g) { if (fscanf(g, "%ld", &period) != 1) period = 0; fclose(g); }
        if (quota > 0 && period > 0) { long c = (quota + period - 1) / period; if (c > 0 && c < n) n = (int)c; }
    }
    if (n < 1) n = 1;
    cached = n;
    return n;
}

static inline int glc_host_cpus(void) {
    int n = glc_host_cpus_hw();
    const int omp = omp_get_max_threads();           /* OMP_NUM_THREADS / omp_set_num_threads: may change at run time */
    if (omp > 0 && omp < n) n = omp;
    return n;
}
#endif
