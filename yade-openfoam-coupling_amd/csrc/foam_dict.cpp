// See foam_dict.hpp.  Host-only code (no HIP): part of libfoamyade_hip.so so that the case reader is reachable through the C-ABI.
#include "foam_dict.hpp"

#include <cctype>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>

namespace fy {

namespace {

struct Tok { std::string s; int line; };

bool tokenize(const std::string& t, std::vector<Tok>* out, std::string* err, const std::string& dir = std::string(), int depth_inc = 0) {
    size_t i = 0, n = t.size();
    int line = 1;
    bool binary = false;
    while (i < n) {
        const char c = t[i];
        if (c == '\n') { ++line; ++i; continue; }
        if (std::isspace((unsigned char)c)) { ++i; continue; }
        if (c == '/' && i + 1 < n && t[i + 1] == '/') { while (i < n && t[i] != '\n') ++i; continue; }
        if (c == '/' && i + 1 < n && t[i + 1] == '*') {
            i += 2;
            while (i + 1 < n && !(t[i] == '*' && t[i + 1] == '/')) { if (t[i] == '\n') ++line; ++i; }
            if (i + 1 >= n) { *err = "unterminated /* comment"; return false; }
            i += 2;
            continue;
        }
        if (c == '{' || c == '}' || c == '(' || c == ')' || c == '[' || c == ']' || c == ';') {
            out->push_back({std::string(1, c), line});
            ++i;
            if (c == ';' && out->size() >= 3 && (*out)[out->size() - 3].s == "format" && (*out)[out->size() - 2].s == "binary") binary = true;      // FoamFile { format binary; }
            if (c == '(' && binary && out->size() >= 3) {
                // binary stream format [OF-6 UList<T>::writeEntry / Istream read of contiguous lists]: `List<scalar> N (` is followed by
                // N * sizeof(T) raw bytes (native doubles: arch "LSB;label=32;scalar=64") and `)`.  The bytes become ONE token, marked
                // by a leading 0x01, which foam_read_list unpacks
                const std::string& ty = (*out)[out->size() - 3].s;
                double cnt;
                const int ncomp = ty == "List<scalar>" ? 1 : ty == "List<vector>" ? 3 : ty == "List<symmTensor>" ? 6 : ty == "List<tensor>" ? 9 : 0;
                if (ncomp && foam_tok_is_number((*out)[out->size() - 2].s, &cnt) && cnt >= 0) {
                    const size_t bytes = (size_t)cnt * (size_t)ncomp * sizeof(double);
                    if (i + bytes > n) { *err = "binary list at line " + std::to_string(line) + " runs past the end of the file"; return false; }
                    std::string blob(1, '\x01');
                    blob.append(t, i, bytes);
                    out->push_back({blob, line});
                    i += bytes;
                }
            }
            continue;
        }
        if (c == '"') {
            size_t j = i + 1;
            while (j < n && t[j] != '"') { if (t[j] == '\n') ++line; ++j; }
            if (j >= n) { *err = "unterminated string at line " + std::to_string(line); return false; }
            out->push_back({t.substr(i + 1, j - i - 1), line});
            i = j + 1;
            continue;
        }
        if (c == '#') {
            // #include "file" / #includeIfPresent "file" (relative to the including file's directory) [OF-6 functionEntries::includeEntry];
            // #inputMode is accepted and ignored (merge is what this reader does); every other directive (#calc, #codeStream, #includeEtc ...) is refused
            size_t j = i + 1;
            while (j < n && std::isalpha((unsigned char)t[j])) ++j;
            const std::string dname = t.substr(i + 1, j - i - 1);
            if (dname == "inputMode") { while (j < n && t[j] != '\n') ++j; i = j; continue; }
            if (dname != "include" && dname != "includeIfPresent") { *err = "directive '#" + dname + "' at line " + std::to_string(line) + " is not supported"; return false; }
            while (j < n && (t[j] == ' ' || t[j] == '\t')) ++j;
            if (j >= n || t[j] != '"') { *err = "#" + dname + " at line " + std::to_string(line) + " needs a quoted file name"; return false; }
            const size_t q = t.find('"', j + 1);
            if (q == std::string::npos) { *err = "unterminated string at line " + std::to_string(line); return false; }
            std::string fname = t.substr(j + 1, q - j - 1);
            if (!fname.empty() && fname[0] != '/' && !dir.empty()) fname = dir + "/" + fname;
            if (depth_inc > 16) { *err = "#include nested too deeply at line " + std::to_string(line); return false; }
            std::ifstream f(fname);
            if (!f) {
                if (dname == "includeIfPresent") { i = q + 1; continue; }
                *err = "#include at line " + std::to_string(line) + ": cannot open " + fname;
                return false;
            }
            std::stringstream ss;
            ss << f.rdbuf();
            const size_t slash = fname.find_last_of('/');
            std::string e2;
            if (!tokenize(ss.str(), out, &e2, slash == std::string::npos ? std::string() : fname.substr(0, slash), depth_inc + 1)) { *err = fname + ": " + e2; return false; }
            i = q + 1;
            continue;
        }
        if (c == '$' && i + 1 < n && t[i + 1] == '{') {                     // ${name}: one token
            const size_t q = t.find('}', i);
            if (q == std::string::npos) { *err = "unterminated ${...} at line " + std::to_string(line); return false; }
            out->push_back({t.substr(i, q - i + 1), line});
            i = q + 1;
            continue;
        }
        // a word; like OpenFOAM's, a word that starts with a letter may carry balanced parentheses: div(phi,U), grad(U), interpolate(HbyA)
        size_t j = i;
        int depth = 0;
        const bool alpha = std::isalpha((unsigned char)c) != 0;
        while (j < n && !std::isspace((unsigned char)t[j]) && t[j] != '{' && t[j] != '}' && t[j] != '[' && t[j] != ']' && t[j] != ';' && t[j] != '"') {
            if (t[j] == '(') { if (!alpha || j == i) break; ++depth; }
            if (t[j] == ')') { if (depth == 0) break; --depth; }
            ++j;
        }
        out->push_back({t.substr(i, j - i), line});
        i = j;
    }
    return true;
}

// $name / ${name}: the entry `name` of this dictionary or of an enclosing one, as read so far [OF-6 primitiveEntry::expandVariable]
const FoamDict::Entry* lookup_macro(const std::string& ref, const FoamDict* d, const std::vector<const FoamDict*>& scopes) {
    std::string name = ref.substr(1);
    if (name.size() >= 2 && name.front() == '{' && name.back() == '}') name = name.substr(1, name.size() - 2);
    if (name.empty() || name.find(':') != std::string::npos || name[0] == '.') return nullptr;      // scoped forms ($:a.b, $..x) are not supported
    auto it = d->e.find(name);
    if (it != d->e.end()) return &it->second;
    for (size_t q = scopes.size(); q-- > 0;) {
        auto jt = scopes[q]->e.find(name);
        if (jt != scopes[q]->e.end()) return &jt->second;
    }
    return nullptr;
}

bool parse_dict(const std::vector<Tok>& tk, size_t* pos, bool top, FoamDict* d, std::string* err, std::vector<const FoamDict*>& scopes) {
    while (*pos < tk.size()) {
        const Tok& k = tk[*pos];
        if (k.s == "}") {
            if (top) { *err = "unexpected '}' at line " + std::to_string(k.line); return false; }
            ++*pos;
            return true;
        }
        if (k.s == ";") { ++*pos; continue; }
        if (k.s == "{" || k.s == ")" || k.s == "]") { *err = "unexpected '" + k.s + "' at line " + std::to_string(k.line); return false; }
        // a top-level bare list (e.g. a field file that is just a list) is not something the case reader needs
        std::string key = k.s;
        ++*pos;
        if (*pos >= tk.size()) { *err = "keyword '" + key + "' at end of file (line " + std::to_string(k.line) + ")"; return false; }
        if (key[0] == '$' && tk[*pos].s == ";") {
            // `$other;` in keyword position: the entries of dictionary `other` are merged in here (later entries override)
            const FoamDict::Entry* m = lookup_macro(key, d, scopes);
            if (!m || !m->sub) { *err = "'" + key + "' at line " + std::to_string(k.line) + " does not name a dictionary read before it"; return false; }
            for (const std::string& kk : m->sub->order) {
                if (!d->e.count(kk)) d->order.push_back(kk);
                d->e[kk] = m->sub->e.at(kk);
            }
            ++*pos;
            continue;
        }
        FoamDict::Entry en;
        if (tk[*pos].s == "{") {
            ++*pos;
            en.sub.reset(new FoamDict());
            scopes.push_back(d);
            const bool ok = parse_dict(tk, pos, false, en.sub.get(), err, scopes);
            scopes.pop_back();
            if (!ok) return false;
        } else {
            int depth = 0;
            for (;;) {
                if (*pos >= tk.size()) { *err = "entry '" + key + "' (line " + std::to_string(k.line) + ") is not terminated by ';'"; return false; }
                const std::string& s = tk[*pos].s;
                if (s == "(" || s == "[") ++depth;
                if (s == ")" || s == "]") --depth;
                if (depth < 0) { *err = "unbalanced ')' in entry '" + key + "' at line " + std::to_string(tk[*pos].line); return false; }
                if (s == "{" && depth > 0) {
                    // dictionaries inside lists (blockMeshDict `boundary ( name { ... } ... )`): keep them as tokens; the consumer
                    // re-parses the slice
                }
                if (s == ";" && depth == 0) { ++*pos; break; }
                if ((s == "{" || s == "}") && depth == 0) { *err = "unexpected '" + s + "' in entry '" + key + "' at line " + std::to_string(tk[*pos].line); return false; }
                if (s.size() > 1 && s[0] == '$') {                    // macro in value position: the referenced entry's tokens (or its dictionary)
                    const FoamDict::Entry* m = lookup_macro(s, d, scopes);
                    if (!m) { *err = "'" + s + "' at line " + std::to_string(tk[*pos].line) + " does not name an entry read before it"; return false; }
                    if (m->sub) {
                        if (!en.tok.empty() || *pos + 1 >= tk.size() || tk[*pos + 1].s != ";") { *err = "'" + s + "' at line " + std::to_string(tk[*pos].line) + " names a dictionary inside a token stream"; return false; }
                        en.sub = m->sub;                                // `key $dict;` : a copy of the dictionary
                        *pos += 2;
                        break;
                    }
                    en.tok.insert(en.tok.end(), m->tok.begin(), m->tok.end());
                    ++*pos;
                    continue;
                }
                en.tok.push_back(s);
                ++*pos;
            }
        }
        if (!d->e.count(key)) d->order.push_back(key);
        d->e[key] = en;
    }
    if (!top) { *err = "missing '}' at end of file"; return false; }
    return true;
}

}  // namespace

const FoamDict* FoamDict::subdict(const std::string& k) const {
    auto it = e.find(k);
    return (it == e.end() || !it->second.sub) ? nullptr : it->second.sub.get();
}
const std::vector<std::string>* FoamDict::tokens(const std::string& k) const {
    auto it = e.find(k);
    return (it == e.end() || it->second.sub) ? nullptr : &it->second.tok;
}

bool foam_tok_is_number(const std::string& t, double* v) {
    if (t.empty()) return false;
    char* end = nullptr;
    const double x = std::strtod(t.c_str(), &end);
    if (end == t.c_str() || *end != '\0') return false;
    if (v) *v = x;
    return true;
}

bool FoamDict::scalar(const std::string& k, double* out) const {
    const auto* t = tokens(k);
    if (!t) return false;
    int depth = 0;
    bool found = false;
    for (const std::string& s : *t) {                 // the last number outside [ ] and ( )
        if (s == "[" || s == "(") { ++depth; continue; }
        if (s == "]" || s == ")") { --depth; continue; }
        double v;
        if (depth == 0 && foam_tok_is_number(s, &v)) { *out = v; found = true; }
    }
    return found;
}
bool FoamDict::integer(const std::string& k, int* out) const {
    double v;
    if (!scalar(k, &v)) return false;
    *out = (int)v;
    return (double)*out == v;
}
bool FoamDict::word(const std::string& k, std::string* out) const {
    const auto* t = tokens(k);
    if (!t || t->empty()) return false;
    *out = (*t)[0];
    return true;
}
bool FoamDict::boolean(const std::string& k, bool* out) const {
    std::string w;
    if (!word(k, &w)) return false;
    if (w == "yes" || w == "on" || w == "true" || w == "y" || w == "t") { *out = true; return true; }
    if (w == "no" || w == "off" || w == "false" || w == "n" || w == "f" || w == "none") { *out = false; return true; }
    return false;
}
bool FoamDict::vector3(const std::string& k, double out[3]) const {
    const auto* t = tokens(k);
    if (!t) return false;
    for (size_t i = 0; i + 4 < t->size() + 1; ++i) {                      // first "( a b c )"
        if ((*t)[i] != "(" || i + 4 >= t->size() || (*t)[i + 4] != ")") continue;
        double v[3];
        if (foam_tok_is_number((*t)[i + 1], &v[0]) && foam_tok_is_number((*t)[i + 2], &v[1]) && foam_tok_is_number((*t)[i + 3], &v[2])) {
            out[0] = v[0]; out[1] = v[1]; out[2] = v[2];
            return true;
        }
    }
    return false;
}

bool foam_read_list(const std::vector<std::string>& tok, size_t i, int ncomp, std::vector<double>* out) {
    double cnt = -1;
    if (i < tok.size() && foam_tok_is_number(tok[i], &cnt)) ++i;         // optional element count
    if (i >= tok.size() || tok[i] != "(") return false;
    ++i;
    out->clear();
    if (i < tok.size() && !tok[i].empty() && tok[i][0] == '\x01') {       // a binary list's bytes (see tokenize)
        const size_t bytes = tok[i].size() - 1;
        if (bytes % sizeof(double) != 0 || i + 1 >= tok.size() || tok[i + 1] != ")") return false;
        out->resize(bytes / sizeof(double));
        if (bytes) std::memcpy(out->data(), tok[i].data() + 1, bytes);
        if (cnt >= 0 && (size_t)cnt * (size_t)ncomp != out->size()) return false;
        return true;
    }
    while (i < tok.size() && tok[i] != ")") {
        if (ncomp == 1) {
            double v;
            if (!foam_tok_is_number(tok[i], &v)) return false;
            out->push_back(v);
            ++i;
        } else {
            if (tok[i] != "(") return false;
            ++i;
            for (int c = 0; c < ncomp; ++c, ++i) {
                double v;
                if (i >= tok.size() || !foam_tok_is_number(tok[i], &v)) return false;
                out->push_back(v);
            }
            if (i >= tok.size() || tok[i] != ")") return false;
            ++i;
        }
    }
    if (i >= tok.size()) return false;
    if (cnt >= 0 && (size_t)cnt * (size_t)ncomp != out->size()) return false;
    return true;
}

bool foam_parse(const std::string& text, FoamDict* out, std::string* err, const std::string& dir) {
    std::vector<Tok> tk;
    if (!tokenize(text, &tk, err, dir)) return false;
    size_t pos = 0;
    std::vector<const FoamDict*> scopes;
    return parse_dict(tk, &pos, true, out, err, scopes);
}

// a file that is a FoamFile header followed by ONE bare list (constant/polyMesh/points, faces, owner, neighbour, boundary): the tokens of the list,
// header skipped
bool foam_list_file_tokens(const std::string& path, std::vector<std::string>* out, std::string* err) {
    std::ifstream f(path, std::ios::binary);
    if (!f) { *err = "cannot open " + path; return false; }
    std::stringstream ss;
    ss << f.rdbuf();
    std::vector<Tok> tk;
    std::string e2;
    if (!tokenize(ss.str(), &tk, &e2)) { *err = path + ": " + e2; return false; }
    size_t i = 0;
    if (i + 1 < tk.size() && tk[i].s == "FoamFile" && tk[i + 1].s == "{") {
        int depth = 0;
        for (; i < tk.size(); ++i) {
            if (tk[i].s == "{") ++depth;
            if (tk[i].s == "}" && --depth == 0) { ++i; break; }
        }
    }
    out->clear();
    out->reserve(tk.size() - i);
    for (; i < tk.size(); ++i) out->push_back(std::move(tk[i].s));
    return true;
}

// The numbers of a file that is a FoamFile header followed by ONE bare list of numbers (constant/polyMesh/points, faces, owner, neighbour), in file
// order, parentheses dropped: points -> N x y z x y z ..., faces -> N 4 a b c d 4 a b c d ..., owner -> N l l l ...  A scanner of its own: these lists have
// tens of millions of entries, and the dictionary lexer above makes a std::string of every token.
template <typename T>
static bool numeric_list_file(const std::string& path, std::vector<T>* out, std::string* err) {
    std::ifstream f(path, std::ios::binary);
    if (!f) { *err = "cannot open " + path; return false; }
    std::stringstream ss;
    ss << f.rdbuf();
    const std::string t = ss.str();
    size_t i = 0;
    const size_t n = t.size();
    auto skip = [&]() {
        for (;;) {
            while (i < n && std::isspace((unsigned char)t[i])) ++i;
            if (i + 1 < n && t[i] == '/' && t[i + 1] == '/') { while (i < n && t[i] != '\n') ++i; continue; }
            if (i + 1 < n && t[i] == '/' && t[i + 1] == '*') { const size_t q = t.find("*/", i + 2); i = q == std::string::npos ? n : q + 2; continue; }
            break;
        }
    };
    skip();
    if (t.compare(i, 8, "FoamFile") == 0) {
        const size_t q = t.find('}', i);
        if (q == std::string::npos) { *err = path + ": unterminated FoamFile header"; return false; }
        {   // the header's `format` ENTRY says so, not the word somewhere in it (a `note "... binary ..."` is text)
            size_t f = i;
            bool is_binary = false;
            while ((f = t.find("format", f)) != std::string::npos && f < q) {
                const bool word = (f == 0 || !(std::isalnum((unsigned char)t[f - 1]) || t[f - 1] == '_')) && f + 6 < n && std::isspace((unsigned char)t[f + 6]);
                size_t v = f + 6;
                while (v < q && std::isspace((unsigned char)t[v])) ++v;
                if (word && t.compare(v, 6, "binary") == 0 && v + 6 < n && (t[v + 6] == ';' || std::isspace((unsigned char)t[v + 6]))) is_binary = true;
                f += 6;
            }
            if (is_binary) {
                // binary stream format [OF-6 UList<T>::writeEntry / List<T>::readList on a binary Istream]: `N(` + the elements as they lie in memory + `)`.
                //   points (vectorField): N (x y z) triples of `scalar`;  owner / neighbour (labelList): N labels;
                //   faces: faceCompactList = TWO lists, the N + 1 offsets into the second one and the point labels of all faces back to back
                // widths from the header's arch "LSB;label=32;scalar=64" (absent: these).  Output in the ASCII scanner's convention (count first; a face as n a b c ...)
                const std::string hdr = t.substr(i, q - i);
                const bool lab64 = hdr.find("label=64") != std::string::npos, sc32 = hdr.find("scalar=32") != std::string::npos;
                const bool compact = hdr.find("faceCompactList") != std::string::npos, vectors = hdr.find("vectorField") != std::string::npos;
                const bool labels = hdr.find("labelList") != std::string::npos;
                if (!compact && !vectors && !labels) { *err = path + ": binary file of a class other than vectorField, labelList, faceCompactList is not supported"; return false; }
                size_t at = q + 1;
                auto read_block = [&](size_t elem_bytes, size_t per, const char** data, size_t* count) -> bool {
                    i = at; skip();
                    char* end = nullptr;
                    const long long cnt = std::strtoll(t.c_str() + i, &end, 10);
                    if (end == t.c_str() + i || cnt < 0) { *err = path + ": binary list without a count"; return false; }
                    i = (size_t)(end - t.c_str());
                    while (i < n && t[i] != '(') ++i;
                    const size_t bytes = (size_t)cnt * per * elem_bytes;
                    if (i >= n || i + 1 + bytes + 1 > n || t[i + 1 + bytes] != ')') { *err = path + ": binary list of " + std::to_string(cnt) + " elements runs past the end of the file (or is not closed)"; return false; }
                    *data = t.data() + i + 1; *count = (size_t)cnt;
                    at = i + 1 + bytes + 1;
                    return true;
                };
                auto label_at = [&](const char* d, size_t k) -> long long {
                    if (lab64) { int64_t v; std::memcpy(&v, d + 8 * k, 8); return (long long)v; }
                    int32_t v; std::memcpy(&v, d + 4 * k, 4); return (long long)v;
                };
                out->clear();
                const char* d = nullptr; size_t cnt = 0;
                if (vectors) {
                    if (!read_block(sc32 ? 4 : 8, 3, &d, &cnt)) return false;
                    out->reserve(3 * cnt + 1);
                    out->push_back((T)cnt);
                    for (size_t k = 0; k < 3 * cnt; ++k) { if (sc32) { float v; std::memcpy(&v, d + 4 * k, 4); out->push_back((T)v); } else { double v; std::memcpy(&v, d + 8 * k, 8); out->push_back((T)v); } }
                } else if (labels) {
                    if (!read_block(lab64 ? 8 : 4, 1, &d, &cnt)) return false;
                    out->reserve(cnt + 1);
                    out->push_back((T)cnt);
                    for (size_t k = 0; k < cnt; ++k) out->push_back((T)label_at(d, k));
                } else {
                    const char* d2 = nullptr; size_t cnt2 = 0;
                    if (!read_block(lab64 ? 8 : 4, 1, &d, &cnt) || cnt < 1 || !read_block(lab64 ? 8 : 4, 1, &d2, &cnt2)) { if (err->empty()) *err = path + ": malformed faceCompactList"; return false; }
                    out->reserve(cnt + cnt2);
                    out->push_back((T)(cnt - 1));
                    for (size_t f = 0; f + 1 < cnt; ++f) {
                        const long long a = label_at(d, f), b = label_at(d, f + 1);
                        if (a < 0 || b < a || (size_t)b > cnt2) { *err = path + ": faceCompactList offsets out of order"; return false; }
                        out->push_back((T)(b - a));
                        for (long long k = a; k < b; ++k) out->push_back((T)label_at(d2, (size_t)k));
                    }
                }
                return true;
            }
        }
        i = q + 1;
    }
    out->clear();
    for (;;) {
        skip();
        if (i >= n) break;
        const char c = t[i];
        if (c == '(' || c == ')') { ++i; continue; }
        char* end = nullptr;
        const double v = std::strtod(t.c_str() + i, &end);
        if (end == t.c_str() + i) { *err = path + ": unexpected '" + std::string(1, c) + "' in a list of numbers"; return false; }
        out->push_back((T)v);
        i = (size_t)(end - t.c_str());
    }
    return true;
}
bool foam_numeric_list_file(const std::string& path, std::vector<double>* out, std::string* err) { return numeric_list_file(path, out, err); }
bool foam_label_list_file(const std::string& path, std::vector<int32_t>* out, std::string* err) { return numeric_list_file(path, out, err); }

bool foam_parse_file(const std::string& path, FoamDict* out, std::string* err) {
    std::ifstream f(path, std::ios::binary);
    if (!f) { *err = "cannot open " + path; return false; }
    std::stringstream ss;
    ss << f.rdbuf();
    std::string e2;
    const size_t slash = path.find_last_of('/');
    if (!foam_parse(ss.str(), out, &e2, slash == std::string::npos ? std::string() : path.substr(0, slash))) { *err = path + ": " + e2; return false; }
    return true;
}

}  // namespace fy
