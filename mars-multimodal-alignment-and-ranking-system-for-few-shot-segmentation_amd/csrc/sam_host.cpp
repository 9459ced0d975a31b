// Host routine of the mask generator: greedy box NMS (what torchvision.ops.batched_nms computes for a
// single category; /root/reference/segment_anything/automatic_mask_generator.py:370-376).
#include <algorithm>
#include <cstdint>
#include <numeric>
#include <vector>

#include "../../include/mars_hip.h"

void mars_set_error(const char* fmt, ...);

extern "C" int marshost_box_nms(const float* boxes, const float* scores, int n, float iou_threshold, int32_t* keep, int32_t* nkeep) {
    if (!boxes || !scores || !keep || !nkeep || n < 0) {
        mars_set_error("box_nms: bad arguments");
        return -1;
    }
    std::vector<int> order(n);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return scores[a] > scores[b]; });
    std::vector<char> dead(n, 0);
    int cnt = 0;
    for (int a = 0; a < n; ++a) {
        const int i = order[a];
        if (dead[i]) continue;
        keep[cnt++] = i;
        const float* bi = boxes + 4 * (int64_t)i;
        const double ai = ((double)bi[2] - bi[0]) * ((double)bi[3] - bi[1]);
        for (int c = a + 1; c < n; ++c) {
            const int j = order[c];
            if (dead[j]) continue;
            const float* bj = boxes + 4 * (int64_t)j;
            const double iw = std::min<double>(bi[2], bj[2]) - std::max<double>(bi[0], bj[0]);
            const double ih = std::min<double>(bi[3], bj[3]) - std::max<double>(bi[1], bj[1]);
            const double inter = std::max(iw, 0.0) * std::max(ih, 0.0);
            const double aj = ((double)bj[2] - bj[0]) * ((double)bj[3] - bj[1]);
            if (inter / (ai + aj - inter) > (double)iou_threshold) dead[j] = 1;
        }
    }
    *nkeep = cnt;
    return 0;
}
