// Host routines of the mask generator: greedy box NMS (what torchvision.ops.batched_nms computes for a
// single category; /root/reference/segment_anything/automatic_mask_generator.py:370-376) and the small-region
// filter (segment_anything/utils/amg.py:274-298).
#include <algorithm>
#include <climits>
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

// 8-connected components of the non-zero pixels, labelled 1.. in raster order of their first pixel; returns the
// number of components and fills `size` (index 0 unused).
static int label_components(const uint8_t* on, int h, int w, std::vector<int32_t>& label, std::vector<int64_t>& size) {
    label.assign((size_t)h * w, 0);
    size.assign(1, 0);
    std::vector<int32_t> stack;
    int n = 0;
    for (int y0 = 0; y0 < h; ++y0)
        for (int x0 = 0; x0 < w; ++x0) {
            const size_t p0 = (size_t)y0 * w + x0;
            if (!on[p0] || label[p0]) continue;
            ++n;
            int64_t cnt = 0;
            label[p0] = n;
            stack.push_back((int32_t)p0);
            while (!stack.empty()) {
                const int32_t p = stack.back();
                stack.pop_back();
                ++cnt;
                const int y = p / w, x = p % w;
                for (int dy = -1; dy <= 1; ++dy)
                    for (int dx = -1; dx <= 1; ++dx) {
                        const int yy = y + dy, xx = x + dx;
                        if (yy < 0 || yy >= h || xx < 0 || xx >= w) continue;
                        const size_t q = (size_t)yy * w + xx;
                        if (on[q] && !label[q]) {
                            label[q] = n;
                            stack.push_back((int32_t)q);
                        }
                    }
            }
            size.push_back(cnt);
        }
    return n;
}

// remove_small_regions (amg.py:274-298), in place on a {0,1} byte mask.  holes != 0: background components
// smaller than area_thresh are filled; holes == 0: foreground components smaller than area_thresh are removed,
// and if every one is, the largest (the first of equals) stays.  *changed = 1 iff a region was below the threshold.
extern "C" int marshost_remove_small_regions(uint8_t* mask, int h, int w, float area_thresh, int holes, int32_t* changed) {
    if (!mask || !changed || h <= 0 || w <= 0 || (int64_t)h * w > INT32_MAX) {
        mars_set_error("remove_small_regions: bad arguments");
        return -1;
    }
    const size_t np = (size_t)h * w;
    std::vector<uint8_t> work(np);
    for (size_t i = 0; i < np; ++i) work[i] = holes ? (mask[i] == 0) : (mask[i] != 0);
    std::vector<int32_t> label;
    std::vector<int64_t> size;
    const int n = label_components(work.data(), h, w, label, size);
    std::vector<char> small(n + 1, 0);
    int nsmall = 0;
    for (int i = 1; i <= n; ++i)
        if ((float)size[i] < area_thresh) { small[i] = 1; ++nsmall; }
    *changed = nsmall > 0;
    if (nsmall == 0) return 0;
    if (holes) {                                    // the mask becomes: background (label 0 of the working image) or a small hole
        for (size_t i = 0; i < np; ++i) mask[i] = label[i] == 0 || small[label[i]];
    } else {
        int keep_one = 0;
        if (nsmall == n) {                          // every region is below the threshold: the largest stays
            keep_one = 1;
            for (int i = 2; i <= n; ++i)
                if (size[i] > size[keep_one]) keep_one = i;
        }
        for (size_t i = 0; i < np; ++i) mask[i] = label[i] != 0 && (!small[label[i]] || label[i] == keep_one);
    }
    return 0;
}
