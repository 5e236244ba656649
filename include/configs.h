/* Defaults of /root/reference/include/configs.h:4-7; every one is also a runtime setting
 * (environment GLICLASS_BATCH_SIZE / GLICLASS_MAX_LENGTH / GLICLASS_THRESHOLD), see host/model.c. */
#ifndef CONFIGS_H
#define CONFIGS_H
#define BATCH_SIZE 8
#define MAX_LENGTH 2048
#define THRESHOLD 0.5f
#define NUM_THREADS 8
#endif
