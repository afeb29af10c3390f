// Probe: do event-record nodes captured into a hipGraph give usable hipEventElapsedTime values, and can
// the events be swapped per replay with hipGraphExecEventRecordNodeSetEvent?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void spin(float* p, int n) { float v = p[threadIdx.x]; for (int i = 0; i < n; i++) v = v * 1.0001f + 0.5f; p[threadIdx.x] = v; }
int main() {
    hipStream_t s; CK(hipStreamCreate(&s));
    float* d; CK(hipMalloc(&d, 4096));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipGraph_t g; hipGraphExec_t ex;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    spin<<<1, 64, 0, s>>>(d, 1000);
    CK(hipEventRecord(a, s));
    spin<<<1, 64, 0, s>>>(d, 200000);
    CK(hipEventRecord(b, s));
    spin<<<1, 64, 0, s>>>(d, 1000);
    CK(hipStreamEndCapture(s, &g));
    size_t nn = 0; CK(hipGraphGetNodes(g, nullptr, &nn));
    std::vector<hipGraphNode_t> nodes(nn); CK(hipGraphGetNodes(g, nodes.data(), &nn));
    hipGraphNode_t na = nullptr, nb = nullptr;
    for (auto n : nodes) {
        hipGraphNodeType t; CK(hipGraphNodeGetType(n, &t));
        if (t == hipGraphNodeTypeEventRecord) { hipEvent_t ev; CK(hipGraphEventRecordNodeGetEvent(n, &ev)); if (ev == a) na = n; if (ev == b) nb = n; }
    }
    printf("nodes %zu, event nodes found: %d %d\n", nn, na != nullptr, nb != nullptr);
    CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ex, s)); CK(hipStreamSynchronize(s));
    float ms = -1; hipError_t e = hipEventElapsedTime(&ms, a, b);
    printf("replay 1: %s %.3f ms\n", hipGetErrorString(e), ms);
    std::vector<hipEvent_t> pa(4), pb(4);
    for (int i = 0; i < 4; i++) { CK(hipEventCreate(&pa[i])); CK(hipEventCreate(&pb[i])); }
    for (int i = 0; i < 4; i++) {
        CK(hipGraphExecEventRecordNodeSetEvent(ex, na, pa[i]));
        CK(hipGraphExecEventRecordNodeSetEvent(ex, nb, pb[i]));
        CK(hipGraphLaunch(ex, s));
    }
    CK(hipStreamSynchronize(s));
    for (int i = 0; i < 4; i++) { ms = -1; e = hipEventElapsedTime(&ms, pa[i], pb[i]); printf("swap %d: %s %.3f ms\n", i, hipGetErrorString(e), ms); }
    e = hipEventElapsedTime(&ms, pa[0], pb[3]); printf("span 0..3: %s %.3f ms\n", hipGetErrorString(e), ms);
    return 0;
}
