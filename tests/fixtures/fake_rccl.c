/* Test double for librccl (tests/test_distributed_cpu.py): records what RcclCommunicator hands to
 * ncclCommInitRank, so that the world_size-2 gloo test can check on the CPU that every rank receives
 * rank 0's unique id whole (128 bytes, NULs included), its own rank and the world size.  No GPU. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct { char internal[128]; } ncclUniqueId;

int ncclGetUniqueId(ncclUniqueId *id) {
  for (int i = 0; i < 128; ++i) id->internal[i] = (char)((i * 37 + 11) & 0xff);
  id->internal[0] = 42; id->internal[1] = 0; id->internal[2] = 0; id->internal[3] = 7;   /* embedded NULs */
  id->internal[64] = 0;
  return 0;
}

int ncclCommInitRank(void **comm, int nranks, ncclUniqueId id, int rank) {
  const char *dir = getenv("FAKE_RCCL_OUT");
  if (!dir) return 5;
  char path[512];
  snprintf(path, sizeof path, "%s/rank%d.bin", dir, rank);
  FILE *f = fopen(path, "wb");
  if (!f) return 5;
  fwrite(&nranks, sizeof nranks, 1, f);
  fwrite(&rank, sizeof rank, 1, f);
  fwrite(id.internal, 1, 128, f);
  fclose(f);
  int *c = malloc(16);
  c[0] = nranks;
  *comm = c;
  return 0;
}

int ncclCommCount(void *comm, int *count) { *count = ((int *)comm)[0]; return 0; }

int ncclAllReduce(const void *s, void *r, size_t n, int dt, int op, void *comm, void *stream) { return 0; }
int ncclCommDestroy(void *comm) { free(comm); return 0; }
