// temporary: symbols implemented later this round
#include "common.h"
extern "C" uint64_t ph_hash_bytes(const void *, uint64_t) { return 0; }
extern "C" int64_t ph_join_count(const ph_join *) { return 0; }
