// stage_pool_tsan.cc -- ThreadSanitizer driver for my-slam_amd/csrc/stage_pool.h: thousands of back-to-back jobs whose sizes
// alternate (a small job followed by a larger one is the case in which a worker that woke up late for the first could, with the
// round-2 pool, run the second job's function with an index taken from the first: an item done twice, the job returning while
// an item still runs, a dead stack lambda called).  Every item must run exactly once, and no job may return before its items.
#include <atomic>
#include <cstdio>
#include <vector>
#include "stage_pool.h"

int main()
{
    StagePool pool(6);
    const int sizes[] = {2, 64, 3, 17, 8, 128, 5, 2, 96, 4};
    long long bad = 0, total = 0;
    for (int rep = 0; rep < 4000; rep++) {
        const int n = sizes[rep % 10];
        std::vector<std::atomic<int>> hits(n);
        for (auto &h : hits) h.store(0);
        std::atomic<int> done{0};
        {
            int scratch[128];                       // stack data only the job of this iteration may touch
            for (int i = 0; i < n; i++) scratch[i] = -1;
            const std::function<void(int)> fn = [&](int i) { scratch[i] = i; hits[i].fetch_add(1); done.fetch_add(1); };
            pool.parallel_for(n, fn);
            if (done.load() != n) bad++;
            for (int i = 0; i < n; i++) if (hits[i].load() != 1 || scratch[i] != i) bad++;
        }
        total += n;
    }
    printf("items %lld bad %lld cpus %d\n", total, bad, StagePool::usable_cpus());
    return bad ? 1 : 0;
}
