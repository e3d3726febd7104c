// Host-side NUMA placement for the frame hand-off (SURVEY.md section 7 "hard parts", section 8e): with N GPUs every
// GPU's pinned frame slots are read by its copy engine at PCIe rate (8 x 14 k FPS x 3.93 MB = 440 GB/s of host reads on a
// full node), so the slots -- and the threads that fill them -- belong on the CPU socket the GPU hangs off.  No libnuma:
// sysfs for the topology, the raw Linux system calls for affinity, memory policy and the page query.
// The reference is single-device (test/yolo_test.cpp:16) and leaves all of this to the OS.
#pragma once

#include <sched.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace irmv {
namespace numa {

// "0-3,8,10-11" -> {0, 1, 2, 3, 8, 10, 11}; anything malformed -> what was parsed up to there (an empty list = no binding)
inline std::vector<int> parse_cpulist(const char *s)
{
    std::vector<int> out;
    if (!s) return out;
    const char *p = s;
    while (*p) {
        while (*p == ' ' || *p == ',' || *p == '\n' || *p == '\t') p++;
        if (!*p) break;
        char *end = nullptr;
        const long a = strtol(p, &end, 10);
        if (end == p || a < 0) break;
        long b = a;
        p = end;
        if (*p == '-') {
            p++;
            b = strtol(p, &end, 10);
            if (end == p || b < a) break;
            p = end;
        }
        if (b - a > 4096) break;
        for (long c = a; c <= b; c++) out.push_back((int)c);
        if (*p && *p != ',' && *p != '\n' && *p != ' ' && *p != '\t') break;
    }
    return out;
}

inline std::vector<int> node_cpus(int node)
{
    if (node < 0) return {};
    char path[96];
    snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
    FILE *f = fopen(path, "r");
    if (!f) return {};
    char buf[4096];
    const size_t n = fread(buf, 1, sizeof buf - 1, f);
    fclose(f);
    buf[n] = 0;
    return parse_cpulist(buf);
}

// Affinity of the calling thread <- the CPUs of `node` (intersected with what the process may use: a container's cpuset).
// Returns false and changes nothing if the node has no usable CPU.
inline bool bind_thread_to_node(int node)
{
    const std::vector<int> cpus = node_cpus(node);
    if (cpus.empty()) return false;
    cpu_set_t allowed, want;
    CPU_ZERO(&allowed);
    CPU_ZERO(&want);
    if (sched_getaffinity(0, sizeof allowed, &allowed) != 0) return false;
    int n = 0;
    for (int c : cpus)
        if (c < CPU_SETSIZE && CPU_ISSET(c, &allowed)) { CPU_SET(c, &want); n++; }
    if (n == 0) return false;
    return sched_setaffinity(0, sizeof want, &want) == 0;
}

// RAII: bind the calling thread to `node` and prefer its memory for the allocations made inside the scope (the first touch
// of a hipHostMallocNumaUser allocation then lands there); restores the previous affinity and the memory policy the thread
// HAD (a process started under numactl --membind / --interleave, or one that called set_mempolicy itself, keeps it;
// MPOL_DEFAULT only where the query failed).
class ScopedNode {
  public:
    explicit ScopedNode(int node)
    {
        if (node < 0) return;
        have_prev_ = sched_getaffinity(0, sizeof prev_, &prev_) == 0;
        bound_ = bind_thread_to_node(node);
        if (node < 1024) {
            have_prev_policy_ = syscall(SYS_get_mempolicy, &prev_mode_, prev_mask_, (unsigned long)(sizeof prev_mask_ * 8), nullptr, 0ul) == 0;
            unsigned long mask[16] = {0};
            mask[node / (8 * sizeof(unsigned long))] |= 1ul << (node % (8 * sizeof(unsigned long)));
            policy_ = syscall(SYS_set_mempolicy, 1 /* MPOL_PREFERRED */, mask, (unsigned long)(sizeof mask * 8)) == 0;
        }
    }
    ~ScopedNode()
    {
        if (policy_) {
            bool restored = false;
            if (have_prev_policy_) {
                bool any = false;
                for (unsigned long w : prev_mask_) any = any || w != 0;
                restored = syscall(SYS_set_mempolicy, prev_mode_, any ? prev_mask_ : nullptr, any ? (unsigned long)(sizeof prev_mask_ * 8) : 0ul) == 0;
            }
            if (!restored) (void)syscall(SYS_set_mempolicy, 0 /* MPOL_DEFAULT */, nullptr, 0ul);
        }
        if (bound_ && have_prev_) (void)sched_setaffinity(0, sizeof prev_, &prev_);
    }
    bool bound() const { return bound_; }
    bool policy() const { return policy_; }
    ScopedNode(const ScopedNode &) = delete;
    ScopedNode &operator=(const ScopedNode &) = delete;

  private:
    cpu_set_t prev_;
    int prev_mode_ = 0;
    unsigned long prev_mask_[16] = {0};
    bool have_prev_ = false, bound_ = false, policy_ = false, have_prev_policy_ = false;
};

// NUMA node that holds the page of `p` (move_pages as a query); < 0: not known (page not present, call not permitted)
inline int page_node(const void *p)
{
    void *page = (void *)((uintptr_t)p & ~(uintptr_t)4095);
    int status = -1;
    const long rc = syscall(SYS_move_pages, 0, 1ul, &page, nullptr, &status, 0);
    return rc == 0 ? status : -1;
}

}  // namespace numa
}  // namespace irmv
