// Version / error plumbing of the C ABI (include/moshi_mi.h).
#include "mmi_common.h"

static thread_local std::string g_last_error;

void mmi_set_error(const std::string& msg) { g_last_error = msg; }
int mmi_fail(int code, const std::string& msg) {
    g_last_error = msg;
    return code;
}

extern "C" int mmi_version(void) { return MMI_ABI_VERSION; }
extern "C" const char* mmi_last_error(void) { return g_last_error.c_str(); }

// ---- launch-list recorder (mmi_graph.h) ------------------------------------------------------------------------------
static thread_local std::vector<std::string>* g_rec_log = nullptr;
static thread_local const char* g_rec_site = "-";
static thread_local long g_rec_bytes = 0;

void mmi_record_begin(std::vector<std::string>* log) { g_rec_log = log; g_rec_site = "-"; }
void mmi_record_site(const char* site) { g_rec_site = site; }
void mmi_record_bytes(long bytes) { g_rec_bytes = bytes; }
void mmi_record_end() { g_rec_log = nullptr; }
void mmi_note_launch(const char* kernel) {
    if (!g_rec_log) return;
    std::string k(kernel);
    while (!k.empty() && (k.front() == '(' || k.front() == ' ')) k.erase(k.begin());      // "(k_gemm_xp<...>)" -> "k_gemm_xp"
    const size_t cut = k.find_first_of("<)");
    if (cut != std::string::npos) k.resize(cut);
    std::string line = std::string(g_rec_site) + "\t" + k;
    if (g_rec_bytes > 0) line += "\t" + std::to_string(g_rec_bytes);
    g_rec_bytes = 0;
    g_rec_log->push_back(line);
}

// copy a launch log ("site\tkernel\n" per launch) into a caller buffer; returns the bytes needed (incl. the final NUL)
int64_t mmi_copy_launch_log(const std::vector<std::string>& log, char* buf, int64_t cap) {
    std::string all;
    for (const auto& l : log) { all += l; all += '\n'; }
    const int64_t need = (int64_t)all.size() + 1;
    if (buf && cap > 0) {
        const int64_t n = need <= cap ? need - 1 : cap - 1;
        memcpy(buf, all.data(), (size_t)n);
        buf[n] = 0;
    }
    return need;
}
