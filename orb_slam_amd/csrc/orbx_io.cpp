// orbx_io.cpp -- detection I/O in the reference's wire format (host only, no GPU).
//
// Frame::ExportToYML (src/Frame.cc:590-609) writes one cv::FileStorage YAML document per frame with
// the keys nNextId, mnId, mTimeStamp, mK, mvKeys, mDescriptors, mvKeysRight, mDescriptorsRight;
// KeyPointLog::ReadFromYAML (include/Util.hpp:400-449) reads mTimeStamp, the two keypoint lists and
// the two descriptor matrices back.  This file writes that layout the way OpenCV 3.4 formats it
// (header "%YAML:1.0", "!!opencv-matrix" maps, integer-valued reals as "31.", other floats as
// "%.8e", doubles as "%.16e", flow sequences wrapped before column 80) and reads it with a
// tolerant scanner that accepts both keypoint layouts OpenCV has used: one flow sequence of 7
// numbers per keypoint inside a block sequence (3.3.1 and later) and a single flat sequence
// (older writers).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/orbx.h"

namespace {

thread_local std::string g_io_error;

int io_fail(int code, const std::string& msg)
{
    g_io_error = msg;
    return code;
}

// cv::FileStorage's number formatting (icvFloatToString / icvDoubleToString)
std::string fmt_real(double v, bool is_float)
{
    char buf[64];
    if (std::isnan(v)) return ".Nan";
    if (std::isinf(v)) return v < 0 ? "-.Inf" : ".Inf";
    if (std::fabs(v) < 2147483647.0 && (double)std::lrint(v) == v) {
        std::snprintf(buf, sizeof(buf), "%ld.", std::lrint(v));
        return buf;
    }
    std::snprintf(buf, sizeof(buf), is_float ? "%.8e" : "%.16e", v);
    return buf;
}

// flow sequence "[ a, b, ... ]" with OpenCV's wrapping: break before an item that would pass column 80
class FlowWriter {
public:
    FlowWriter(FILE* f, int col, int cont_indent) : f_(f), col_(col), indent_(cont_indent) { put("["); }
    void item(const std::string& s)
    {
        if (!first_) put(",");
        first_ = false;
        if (col_ + 1 + (int)s.size() + 1 > 80) {
            std::fputc('\n', f_);
            for (int i = 0; i < indent_; i++) std::fputc(' ', f_);
            col_ = indent_;
        } else {
            put(" ");
        }
        put(s);
    }
    void close() { put(first_ ? "]" : " ]"); std::fputc('\n', f_); }

private:
    void put(const std::string& s) { std::fputs(s.c_str(), f_); col_ += (int)s.size(); }
    FILE* f_;
    int col_, indent_;
    bool first_ = true;
};

void write_keypoints(FILE* f, const char* key, const orbx_keypoint* k, int n, bool flat)
{
    if (flat) {
        std::fprintf(f, "%s: ", key);
        FlowWriter w(f, (int)std::strlen(key) + 2, 4);
        for (int i = 0; i < n; i++) {
            w.item(fmt_real(k[i].x, true)); w.item(fmt_real(k[i].y, true)); w.item(fmt_real(k[i].size, true));
            w.item(fmt_real(k[i].angle, true)); w.item(fmt_real(k[i].response, true));
            w.item(std::to_string(k[i].octave)); w.item(std::to_string(k[i].class_id));
        }
        w.close();
        return;
    }
    if (n == 0) { std::fprintf(f, "%s: []\n", key); return; }
    std::fprintf(f, "%s:\n", key);
    for (int i = 0; i < n; i++) {
        std::fputs("   - ", f);
        FlowWriter w(f, 5, 7);
        w.item(fmt_real(k[i].x, true)); w.item(fmt_real(k[i].y, true)); w.item(fmt_real(k[i].size, true));
        w.item(fmt_real(k[i].angle, true)); w.item(fmt_real(k[i].response, true));
        w.item(std::to_string(k[i].octave)); w.item(std::to_string(k[i].class_id));
        w.close();
    }
}

void write_matrix_u8(FILE* f, const char* key, const uint8_t* d, int rows, int cols)
{
    std::fprintf(f, "%s: !!opencv-matrix\n   rows: %d\n   cols: %d\n   dt: u\n   data: ", key, rows, cols);
    FlowWriter w(f, 9, 7);
    for (long i = 0; i < (long)rows * cols; i++) w.item(std::to_string((int)d[i]));
    w.close();
}

void write_matrix_f32(FILE* f, const char* key, const float* d, int rows, int cols)
{
    std::fprintf(f, "%s: !!opencv-matrix\n   rows: %d\n   cols: %d\n   dt: f\n   data: ", key, rows, cols);
    FlowWriter w(f, 9, 7);
    for (int i = 0; i < rows * cols; i++) w.item(fmt_real(d[i], true));
    w.close();
}

// ---- reader ---------------------------------------------------------------------------------
struct Node {
    std::string scalar;             // "key: value" on one line
    std::vector<double> numbers;    // every number of a sequence / of a matrix's data
    int rows = -1, cols = -1;
    std::string dt;
    bool is_matrix = false, present = false;
};

bool parse_number(const std::string& t, double& v)
{
    if (t == ".Nan" || t == ".NaN" || t == ".nan") { v = std::nan(""); return true; }
    if (t == ".Inf" || t == ".inf") { v = INFINITY; return true; }
    if (t == "-.Inf" || t == "-.inf") { v = -INFINITY; return true; }
    char* end = nullptr;
    v = std::strtod(t.c_str(), &end);
    return end != t.c_str() && *end == '\0';
}

// splits the text after "key:" (possibly many lines) into number tokens, ignoring [ ] , and "- "
void collect_numbers(const std::string& text, std::vector<double>& out)
{
    std::string tok;
    auto flush = [&]() {
        if (tok.empty()) return;
        double v;
        if (parse_number(tok, v)) out.push_back(v);
        tok.clear();
    };
    for (size_t i = 0; i < text.size(); i++) {
        const char ch = text[i];
        const bool list_dash = ch == '-' && tok.empty() && (i + 1 >= text.size() || text[i + 1] == ' ' || text[i + 1] == '\n');
        if (ch == '[' || ch == ']' || ch == ',' || ch == ' ' || ch == '\n' || ch == '\r' || ch == '\t' || list_dash) flush();
        else tok.push_back(ch);
    }
    flush();
}

int top_level_key(const std::string& line, std::string& key, std::string& rest)
{
    if (line.empty() || line[0] == ' ' || line[0] == '-' || line[0] == '%' || line[0] == '#') return 0;
    const size_t c = line.find(':');
    if (c == std::string::npos) return 0;
    key = line.substr(0, c);
    rest = line.substr(c + 1);
    return 1;
}

void finish_node(Node& n, const std::string& body)
{
    n.present = true;
    if (body.find("!!opencv-matrix") != std::string::npos) {
        n.is_matrix = true;
        auto field = [&](const char* name) -> std::string {
            const size_t p = body.find(name);
            if (p == std::string::npos) return "";
            size_t b = p + std::strlen(name), e = body.find('\n', b);
            std::string v = body.substr(b, e == std::string::npos ? std::string::npos : e - b);
            while (!v.empty() && (v[0] == ' ' || v[0] == '"')) v.erase(0, 1);
            while (!v.empty() && (v.back() == ' ' || v.back() == '\r' || v.back() == '"')) v.pop_back();
            return v;
        };
        n.rows = std::atoi(field("rows:").c_str());
        n.cols = std::atoi(field("cols:").c_str());
        n.dt = field("dt:");
        const size_t d = body.find("data:");
        if (d != std::string::npos) collect_numbers(body.substr(d + 5), n.numbers);
        return;
    }
    std::string s = body;
    while (!s.empty() && (s[0] == ' ')) s.erase(0, 1);
    const size_t nl = s.find('\n');
    n.scalar = s.substr(0, nl);
    while (!n.scalar.empty() && (n.scalar.back() == '\r' || n.scalar.back() == ' ')) n.scalar.pop_back();
    collect_numbers(body, n.numbers);
}

// A number of the file as an integer of the target's range: rounded half to even and saturated, as OpenCV's reader
// converts (cvRound + saturate_cast); a plain cast of a value outside the range is undefined.  NaN reads as 0.
long to_int_saturated(double v, long lo, long hi)
{
    if (!(v == v)) return 0;
    if (v <= (double)lo) return lo;
    if (v >= (double)hi) return hi;
    return std::lrint(v);
}

int keypoints_from(const Node& n, orbx_keypoint* out, int cap, int* count, const char* what)
{
    if (!n.present) return io_fail(ORBX_ERR_INVALID, std::string("key missing: ") + what);
    if (n.numbers.size() % 7 != 0) return io_fail(ORBX_ERR_INVALID, std::string(what) + ": not a multiple of 7 numbers");
    const int cnt = (int)(n.numbers.size() / 7);
    *count = cnt;
    if (!out) return ORBX_OK;
    if (cnt > cap) return io_fail(ORBX_ERR_CAPACITY, std::string(what) + ": buffer too small");
    for (int i = 0; i < cnt; i++) {
        const double* v = &n.numbers[(size_t)i * 7];
        out[i].x = (float)v[0]; out[i].y = (float)v[1]; out[i].size = (float)v[2]; out[i].angle = (float)v[3];
        out[i].response = (float)v[4]; out[i].octave = (int32_t)to_int_saturated(v[5], INT32_MIN, INT32_MAX);
        out[i].class_id = (int32_t)to_int_saturated(v[6], INT32_MIN, INT32_MAX);
    }
    return ORBX_OK;
}

int descriptors_from(const Node& n, uint8_t* out, int cap, int count, const char* what)
{
    if (!n.present) return io_fail(ORBX_ERR_INVALID, std::string("key missing: ") + what);
    if (count == 0) return ORBX_OK;    // an empty cv::Mat is written without data
    if (!n.is_matrix || n.dt != "u" || n.cols != 32 || n.rows != count || (long)n.numbers.size() != 32L * count)
        return io_fail(ORBX_ERR_INVALID, std::string(what) + ": expected a " + std::to_string(count) + " x 32 matrix of dt u");
    if (!out) return ORBX_OK;
    if (count > cap) return io_fail(ORBX_ERR_CAPACITY, std::string(what) + ": buffer too small");
    for (long i = 0; i < 32L * count; i++) out[i] = (uint8_t)to_int_saturated(n.numbers[(size_t)i], 0, 255);
    return ORBX_OK;
}

} // namespace

extern "C" {

const char* orbx_io_last_error(void) { return g_io_error.c_str(); }

int orbx_detection_write_yaml(const char* path, double nNextId, double mnId, double mTimeStamp, const float* mK,
                              const orbx_keypoint* keys, const uint8_t* desc, int n,
                              const orbx_keypoint* keys_right, const uint8_t* desc_right, int n_right, int legacy_flat)
{
    if (!path || n < 0 || n_right < 0 || (n > 0 && (!keys || !desc)) || (n_right > 0 && (!keys_right || !desc_right)))
        return io_fail(ORBX_ERR_INVALID, "bad arguments");
    FILE* f = std::fopen(path, "w");
    if (!f) return io_fail(ORBX_ERR_IO, std::string("cannot open for writing: ") + path);
    std::fputs("%YAML:1.0\n---\n", f);
    std::fprintf(f, "nNextId: %s\n", fmt_real(nNextId, false).c_str());
    std::fprintf(f, "mnId: %s\n", fmt_real(mnId, false).c_str());
    std::fprintf(f, "mTimeStamp: %s\n", fmt_real(mTimeStamp, false).c_str());
    if (mK) write_matrix_f32(f, "mK", mK, 3, 3);
    write_keypoints(f, "mvKeys", keys, n, legacy_flat != 0);
    write_matrix_u8(f, "mDescriptors", desc, n, n ? 32 : 0);
    write_keypoints(f, "mvKeysRight", keys_right, n_right, legacy_flat != 0);
    write_matrix_u8(f, "mDescriptorsRight", desc_right, n_right, n_right ? 32 : 0);
    const bool bad = std::ferror(f) != 0;
    if (std::fclose(f) != 0 || bad) return io_fail(ORBX_ERR_IO, std::string("write failed: ") + path);
    return ORBX_OK;
}

int orbx_detection_read_yaml(const char* path, double* mTimeStamp, orbx_keypoint* keys, uint8_t* desc, int cap, int* n,
                             orbx_keypoint* keys_right, uint8_t* desc_right, int cap_right, int* n_right)
{
    if (!path || !mTimeStamp || !n || !n_right) return io_fail(ORBX_ERR_INVALID, "bad arguments");
    *mTimeStamp = -1;   // the reference's "file not found" state (include/Util.hpp:407-413)
    *n = 0;
    *n_right = 0;
    FILE* f = std::fopen(path, "r");
    if (!f) return io_fail(ORBX_ERR_IO, std::string("keypoint file not found: ") + path);
    std::string text;
    char buf[1 << 16];
    size_t got;
    while ((got = std::fread(buf, 1, sizeof(buf), f)) > 0) text.append(buf, got);
    std::fclose(f);

    Node ts, kl, dl, kr, dr;
    std::string cur_key, cur_body;
    auto close_node = [&]() {
        Node* t = cur_key == "mTimeStamp" ? &ts : cur_key == "mvKeys" ? &kl : cur_key == "mDescriptors" ? &dl
                : cur_key == "mvKeysRight" ? &kr : cur_key == "mDescriptorsRight" ? &dr : nullptr;
        if (t) finish_node(*t, cur_body);
    };
    size_t pos = 0;
    while (pos <= text.size()) {
        size_t e = text.find('\n', pos);
        if (e == std::string::npos) e = text.size();
        const std::string line = text.substr(pos, e - pos);
        std::string key, rest;
        if (line == "---" || line == "...") { /* document markers */ }
        else if (top_level_key(line, key, rest)) {
            close_node();
            cur_key = key;
            cur_body = rest + "\n";
        } else {
            cur_body += line + "\n";
        }
        if (e == text.size()) break;
        pos = e + 1;
    }
    close_node();

    if (!ts.present || ts.numbers.size() != 1) return io_fail(ORBX_ERR_INVALID, "mTimeStamp missing or not a number");
    int rc = keypoints_from(kl, keys, cap, n, "mvKeys");
    if (rc != ORBX_OK) return rc;
    rc = descriptors_from(dl, desc, cap, *n, "mDescriptors");
    if (rc != ORBX_OK) return rc;
    rc = keypoints_from(kr, keys_right, cap_right, n_right, "mvKeysRight");
    if (rc != ORBX_OK) return rc;
    rc = descriptors_from(dr, desc_right, cap_right, *n_right, "mDescriptorsRight");
    if (rc != ORBX_OK) return rc;
    *mTimeStamp = ts.numbers[0];
    return ORBX_OK;
}

} // extern "C"
