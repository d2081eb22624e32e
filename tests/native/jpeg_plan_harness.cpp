// Sanitizer harness for the host half of the device JPEG decoder (mvlm_amd/csrc/jpeg_plan.cpp: header parser, Huffman
// table builder, un-stuffing into the staging buffer).  Built by tests/test_jpeg_plan_sanitized.py with
// g++ -fsanitize=address,undefined.  Input: one pack file = { u32 length, bytes } per JPEG.  Every input is copied into a heap
// block of EXACTLY its length and staged into a heap block of EXACTLY mvlm_jpeg_stage_bytes, so a read or write one byte
// outside either is an ASan report.  One line per input: "<index> rc=<plan> [<fill> w h c n_sub used bound]".
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../mvlm_amd/csrc/jpeg_plan.h"

int main(int argc, char** argv) {
    if (argc != 2) return 2;
    FILE* f = std::fopen(argv[1], "rb");
    if (!f) return 2;
    long bad = 0;
    for (int index = 0;; ++index) {
        uint32_t len = 0;
        if (std::fread(&len, 4, 1, f) != 1) break;
        uint8_t* data = static_cast<uint8_t*>(std::malloc(len ? len : 1));
        if (len && std::fread(data, 1, len, f) != len) return 3;
        MvlmJpegPlan* plan = mvlm_jpeg_plan_new();
        std::string why;
        const int rc = mvlm_jpeg_plan_impl(data, len, *plan, why);
        if (rc != 0) {
            std::printf("%d rc=%d %s\n", index, rc, why.c_str());
            if (rc != 2 || why.empty()) ++bad;
        } else {
            int w = 0, h = 0, c = 0;
            mvlm_jpeg_plan_dims(*plan, &w, &h, &c);
            const size_t bound = mvlm_jpeg_stage_bytes(*plan, len);
            uint8_t* stage = static_cast<uint8_t*>(std::malloc(bound));
            const int rc2 = mvlm_jpeg_fill_stage(*plan, data, len, stage, why);
            if (rc2 == 0) {
                const auto* hdr = reinterpret_cast<const mvlm_jpeg::JpegDev*>(stage);
                const auto* sub_seg = reinterpret_cast<const int32_t*>(stage + plan->off_subseg);
                bool ok = plan->stage_used <= bound && hdr->n_sub == plan->dev.n_sub && hdr->n_sub > 0 &&
                          size_t(hdr->n_sub + 2) * mvlm_jpeg::SUB_BYTES == plan->stream_bytes && hdr->width == w && hdr->height == h &&
                          w > 0 && h > 0 && w <= 16384 && h <= 16384 && (c == 1 || c == 3) && hdr->total_blocks > 0 &&
                          size_t(hdr->total_blocks) <= (len - plan->scan_begin) * 4;
                for (int s = 0; ok && s < hdr->n_sub; ++s) ok = sub_seg[s] >= 0 && sub_seg[s] < plan->n_seg && (s == 0 || sub_seg[s] >= sub_seg[s - 1]);
                if (!ok) ++bad;
                std::printf("%d rc=0 fill=0 %d %d %d %d %zu %zu%s\n", index, w, h, c, hdr->n_sub, plan->stage_used, bound, ok ? "" : " INVARIANT");
            } else {
                std::printf("%d rc=0 fill=%d %s\n", index, rc2, why.c_str());
                if (rc2 != 2 || why.empty()) ++bad;
            }
            std::free(stage);
        }
        mvlm_jpeg_plan_delete(plan);
        std::free(data);
    }
    std::fclose(f);
    return bad ? 1 : 0;
}
