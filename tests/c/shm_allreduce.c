/* shm_allreduce.c -- TEST INFRASTRUCTURE: a host-side all-reduce (sum of doubles) between the
 * ranks of one node through a file mapped MAP_SHARED, for the sharded tests of the abstract-vector
 * flavour (this image has no MPI).  It has the signature of nka_hip_host_allreduce_fn
 * (include/nka_hip.h) so that it can be installed with nka_hip_vec_set_host_allreduce /
 * hip_block_vector_set_host_allreduce exactly where a real caller would install a wrapper of
 * MPI_Allreduce.  Every rank adds the contributions in rank order, so all ranks read the same bits.
 *
 * The test harness creates the file (zero-filled, shm_ar_bytes(world) long) before the ranks start.
 */
#include <fcntl.h>
#include <sched.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>

#define SHM_AR_MAXCOUNT 64

typedef struct {
  uint32_t arrived;      /* ranks inside the current barrier */
  uint32_t generation;   /* completed barriers */
  uint32_t calls;        /* all-reduces completed (diagnostic) */
  uint32_t pad[13];
  double slot[1];        /* world * SHM_AR_MAXCOUNT doubles */
} shm_ar_area;

typedef struct {
  int world, rank;
  shm_ar_area *area;
  double timeout_s;
} shm_ar;

size_t shm_ar_bytes(int world) { return sizeof(shm_ar_area) + sizeof(double) * (size_t)world * SHM_AR_MAXCOUNT; }

static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* returns a context for shm_allreduce, or NULL */
void *shm_ar_open(const char *path, int world, int rank) {
  if (world < 1 || rank < 0 || rank >= world) return NULL;
  int fd = open(path, O_RDWR);
  if (fd < 0) return NULL;
  void *p = mmap(NULL, shm_ar_bytes(world), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) return NULL;
  shm_ar *c = (shm_ar *)calloc(1, sizeof *c);
  c->world = world;
  c->rank = rank;
  c->area = (shm_ar_area *)p;
  c->timeout_s = 120.0;
  return c;
}

static int barrier(shm_ar *c) {
  shm_ar_area *a = c->area;
  const uint32_t gen = __atomic_load_n(&a->generation, __ATOMIC_ACQUIRE);
  if (__atomic_add_fetch(&a->arrived, 1u, __ATOMIC_ACQ_REL) == (uint32_t)c->world) {
    __atomic_store_n(&a->arrived, 0u, __ATOMIC_RELEASE);
    __atomic_add_fetch(&a->generation, 1u, __ATOMIC_ACQ_REL);
    return 0;
  }
  const double t0 = now_s();
  while (__atomic_load_n(&a->generation, __ATOMIC_ACQUIRE) == gen) {
    sched_yield();
    if (now_s() - t0 > c->timeout_s) return 1;   /* a peer died: fail instead of hanging the test */
  }
  return 0;
}

/* nka_hip_host_allreduce_fn */
int shm_allreduce(void *ctx, double *vals, int32_t count) {
  shm_ar *c = (shm_ar *)ctx;
  if (!c || count < 0 || count > SHM_AR_MAXCOUNT) return 1;
  shm_ar_area *a = c->area;
  for (int i = 0; i < count; i++) a->slot[(size_t)c->rank * SHM_AR_MAXCOUNT + i] = vals[i];
  if (barrier(c)) return 2;
  for (int i = 0; i < count; i++) {
    double s = 0.0;
    for (int r = 0; r < c->world; r++) s += a->slot[(size_t)r * SHM_AR_MAXCOUNT + i];   /* rank order: same bits everywhere */
    vals[i] = s;
  }
  if (barrier(c)) return 2;    /* nobody overwrites a slot another rank is still reading */
  if (c->rank == 0) __atomic_add_fetch(&a->calls, 1u, __ATOMIC_RELAXED);
  return 0;
}

uint32_t shm_ar_calls(void *ctx) { return ctx ? __atomic_load_n(&((shm_ar *)ctx)->area->calls, __ATOMIC_RELAXED) : 0u; }

/* All-gather of `nbytes` (<= 512) opaque bytes per rank, in rank order -- where a real caller would use MPI_Allgather: the
 * hipIpc handles of the peer-to-peer exchange (nka_hip_p2p_export / _attach), which are NOT numbers and must not be summed. */
int shm_allgather(void *ctx, const void *mine, int32_t nbytes, void *all) {
  shm_ar *c = (shm_ar *)ctx;
  if (!c || nbytes < 0 || (size_t)nbytes > sizeof(double) * SHM_AR_MAXCOUNT) return 1;
  shm_ar_area *a = c->area;
  const char *src = (const char *)mine;
  char *slot = (char *)&a->slot[(size_t)c->rank * SHM_AR_MAXCOUNT];
  for (int i = 0; i < nbytes; i++) slot[i] = src[i];
  if (barrier(c)) return 2;
  for (int r = 0; r < c->world; r++) {
    const char *s = (const char *)&a->slot[(size_t)r * SHM_AR_MAXCOUNT];
    for (int i = 0; i < nbytes; i++) ((char *)all)[(size_t)r * nbytes + i] = s[i];
  }
  if (barrier(c)) return 2;
  return 0;
}

/* a plain barrier of the ranks (before a rank frees memory its peers may still write into) */
int shm_barrier(void *ctx) { return ctx ? barrier((shm_ar *)ctx) : 1; }
