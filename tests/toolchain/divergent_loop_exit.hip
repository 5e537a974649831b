// Reproducer for a hipcc 7.2 (ROCm 7.2.0, gfx950) code-generation fault that librsx works around by construction.
//
// A per-lane loop over an explicit stack — the iterative form of CSG contains() (source_amd/csrc/dev_csg.hpp: node_contains) — written
// in two ways that are the same program:
//   early_exit : `for (;;)` with `continue` after each case and `return result` in the middle of the body
//   single_exit: `while (!finished)` with if / else-if / else, one exit test, no continue, no return inside
// Both are right for a single lane. When the 64 lanes of a wave leave the loop at different turns the early_exit form returned wrong
// answers in librsx's kernels (round 3: 1687 of 20000 points of a nine-level Union chain; traced with printf — a lane alone gets it
// right); round 2 met the same fault in the volume enumeration of k_render_trace_path and held it together with an atomic.
// The program evaluates random Boolean trees per lane with both forms and compares them with the host's recursion.
//   exit code 0: the single_exit form (the one librsx uses) is right; prints whether the early_exit form is too
//   exit code 1: the single_exit form is wrong — the workaround no longer works on this toolchain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

struct Node { int32_t type, a, b, value; };          // type 0 leaf(value), 1 union, 2 intersect, 3 subtract; gate folded into value
#define STACK 64

__device__ __noinline__ bool leaf_value(const Node *nodes, int32_t idx, uint32_t salt) {
    uint32_t h = (uint32_t)nodes[idx].value * 2654435761u ^ salt * 40503u;
    h ^= h >> 13; h *= 2246822519u; h ^= h >> 16;
    return (h & 3u) != 0u;                            // per-lane: three quarters true
}

__device__ __noinline__ bool eval_early_exit(const Node *nodes, int32_t top, uint32_t salt) {
    int32_t frames[STACK];
    int sp = 0;
    int32_t cur = top;
    bool result = false, have = false;
    for (;;) {
        if (!have) {
            const Node n = nodes[cur];
            if (n.type == 0) { result = leaf_value(nodes, cur, salt); have = true; }
            else { frames[sp++] = cur << 1; cur = n.a; }
            continue;
        }
        if (sp == 0) return result;
        const int32_t fr = frames[sp - 1];
        const Node n = nodes[fr >> 1];
        if (!(fr & 1)) {
            const bool need_b = n.type == 1 ? !result : result;
            if (!need_b) { --sp; continue; }
            frames[sp - 1] = fr | 1;
            cur = n.b;
            have = false;
            continue;
        }
        if (n.type == 3) result = !result;
        --sp;
    }
}

__device__ __noinline__ bool eval_single_exit(const Node *nodes, int32_t top, uint32_t salt) {
    int32_t frames[STACK];
    int sp = 0;
    int32_t cur = top;
    bool result = false, have = false, finished = false;
    while (!finished) {
        if (!have) {
            const Node n = nodes[cur];
            if (n.type == 0) { result = leaf_value(nodes, cur, salt); have = true; }
            else { frames[sp] = cur << 1; sp += 1; cur = n.a; }
        } else if (sp == 0) {
            finished = true;
        } else {
            const int32_t fr = frames[sp - 1];
            const Node n = nodes[fr >> 1];
            if (!(fr & 1)) {
                const bool need_b = n.type == 1 ? !result : result;
                if (need_b) { frames[sp - 1] = fr | 1; cur = n.b; have = false; }
                else sp -= 1;
            } else {
                if (n.type == 3) result = !result;
                sp -= 1;
            }
        }
    }
    return result;
}

__global__ void k_eval(const Node *nodes, const int32_t *tops, int n_tops, int n, uint8_t *early, uint8_t *single) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t top = tops[i % n_tops];             // neighbouring lanes walk different trees: they leave the loop at different turns
    early[i] = eval_early_exit(nodes, top, (uint32_t)i) ? 1 : 0;
    single[i] = eval_single_exit(nodes, top, (uint32_t)i) ? 1 : 0;
}

static uint32_t rng_state = 12345u;
static uint32_t rnd() { rng_state = rng_state * 1664525u + 1013904223u; return rng_state >> 8; }
static int build(std::vector<Node> &nodes, int depth) {
    const int id = (int)nodes.size();
    nodes.push_back(Node{0, -1, -1, (int32_t)rnd()});
    if (depth > 0 && rnd() % 10 < 8) {
        nodes[id].type = 1 + (int)(rnd() % 3);
        const int a = build(nodes, depth - 1), b = build(nodes, depth - 1 - (int)(rnd() % 2 ? 0 : (depth > 1)));
        nodes[id].a = a; nodes[id].b = b;
    }
    return id;
}
static bool host_leaf(const std::vector<Node> &nodes, int idx, uint32_t salt) {
    uint32_t h = (uint32_t)nodes[idx].value * 2654435761u ^ salt * 40503u;
    h ^= h >> 13; h *= 2246822519u; h ^= h >> 16;
    return (h & 3u) != 0u;
}
static bool host_eval(const std::vector<Node> &nodes, int idx, uint32_t salt) {
    const Node &n = nodes[idx];
    if (n.type == 0) return host_leaf(nodes, idx, salt);
    const bool a = host_eval(nodes, n.a, salt);
    if (n.type == 1) return a || host_eval(nodes, n.b, salt);
    if (n.type == 2) return a && host_eval(nodes, n.b, salt);
    return a && !host_eval(nodes, n.b, salt);
}

int main() {
    std::vector<Node> nodes;
    std::vector<int32_t> tops;
    for (int t = 0; t < 97; ++t) tops.push_back(build(nodes, 3 + t % 9));
    const int n = 1 << 16;
    Node *d_nodes; int32_t *d_tops; uint8_t *d_early, *d_single;
    if (hipMalloc(&d_nodes, nodes.size() * sizeof(Node)) != hipSuccess) { std::printf("no device\n"); return 2; }
    hipMalloc(&d_tops, tops.size() * 4); hipMalloc(&d_early, n); hipMalloc(&d_single, n);
    hipMemcpy(d_nodes, nodes.data(), nodes.size() * sizeof(Node), hipMemcpyHostToDevice);
    hipMemcpy(d_tops, tops.data(), tops.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_eval, dim3(n / 256), dim3(256), 0, 0, d_nodes, d_tops, (int)tops.size(), n, d_early, d_single);
    std::vector<uint8_t> early(n), single(n);
    hipMemcpy(early.data(), d_early, n, hipMemcpyDeviceToHost);
    hipMemcpy(single.data(), d_single, n, hipMemcpyDeviceToHost);
    int bad_early = 0, bad_single = 0;
    for (int i = 0; i < n; ++i) {
        const bool ref = host_eval(nodes, tops[i % tops.size()], (uint32_t)i);
        bad_early += (early[i] != 0) != ref;
        bad_single += (single[i] != 0) != ref;
    }
    std::printf("early_exit wrong: %d of %d   single_exit wrong: %d of %d\n", bad_early, n, bad_single, n);
    return bad_single ? 1 : 0;
}
