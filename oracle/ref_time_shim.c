/* TEST INFRASTRUCTURE (oracle/): deterministic clock for the compiled reference.
 *
 * The reference seeds both of every player's generators from the wall clock
 * (/root/reference/environment/game_backend/source/PythonHandle.cpp:68-71,
 *  `seed1 = time(NULL), seed2 = time(NULL)`).  This file is linked INTO the
 * reference extension (with -Wl,-Bsymbolic-functions) so that its `time()`
 * call resolves here; nothing in the reference sources is touched or copied.
 * `oracle_set_time(v)` picks the value the next reset()/init()/copy() will see.
 */
#include <time.h>

static long long g_oracle_time = 1000;

__attribute__((visibility("default"))) void oracle_set_time(long long v) { g_oracle_time = v; }
__attribute__((visibility("default"))) long long oracle_get_time(void) { return g_oracle_time; }

__attribute__((visibility("default"))) time_t time(time_t *out) {
    if (out) *out = (time_t)g_oracle_time;
    return (time_t)g_oracle_time;
}
