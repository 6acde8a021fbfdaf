// Minimal reader for OpenFOAM's dictionary file format [OF-6 file format; nothing of it is in /root/reference, which only consumes
// such files through OpenFOAM's IOdictionary / GeometricField constructors, icoFoamYade/createFields.H:3-58,
// pimpleFoamYade/createFields.H:3-110].  Supported: // and /* */ comments, `key tokens...;`, `key { ... }` sub-dictionaries,
// ( ... ) lists (nested, with or without a leading count), [ ... ] dimension sets, quoted strings, the FoamFile header (kept as an
// ordinary sub-dictionary), #include / #includeIfPresent "file" (relative to the including file), $name / ${name} macros (the entry of this or an
// enclosing dictionary as read so far; `$dict;` in keyword position merges a dictionary).  Not supported: #calc / #codeStream / #includeEtc,
// scoped macros ($:a.b, $..x), binary format, regular-expression keys.
#pragma once
#include <cstddef>
#include <map>
#include <memory>
#include <cstdint>
#include <string>
#include <vector>

namespace fy {

struct FoamDict {
    // an entry is either a token stream (everything between the keyword and the ';', parentheses kept as tokens) or a sub-dictionary
    struct Entry {
        std::vector<std::string> tok;
        std::shared_ptr<FoamDict> sub;
    };
    std::vector<std::string> order;                 // keywords in file order (patch order matters)
    std::map<std::string, Entry> e;

    bool has(const std::string& k) const { return e.count(k) != 0; }
    const FoamDict* subdict(const std::string& k) const;
    const std::vector<std::string>* tokens(const std::string& k) const;
    // last numeric token of the entry: handles `nu 0.01;`, `nu [0 2 -1 0 0 0 0] 0.01;` and `nu nu [0 2 -1 0 0 0 0] 0.01;`
    bool scalar(const std::string& k, double* out) const;
    bool integer(const std::string& k, int* out) const;
    bool word(const std::string& k, std::string* out) const;          // first token
    bool boolean(const std::string& k, bool* out) const;              // yes/no/on/off/true/false
    // `( x y z )` anywhere in the entry (the first parenthesised triple), e.g. `value (0 0 -9.81);` or `value uniform (1 0 0);`
    bool vector3(const std::string& k, double out[3]) const;
};

// Parses `text`; on failure returns false and sets *err (with a line number).
bool foam_parse(const std::string& text, FoamDict* out, std::string* err, const std::string& dir = std::string());     // dir: where #include looks
bool foam_parse_file(const std::string& path, FoamDict* out, std::string* err);
bool foam_list_file_tokens(const std::string& path, std::vector<std::string>* out, std::string* err);      // FoamFile header + one bare list: the list's tokens
// FoamFile header + one bare list of numbers, parentheses dropped (polyMesh points / faces / owner / neighbour: tens of millions of entries)
bool foam_numeric_list_file(const std::string& path, std::vector<double>* out, std::string* err);
bool foam_label_list_file(const std::string& path, std::vector<int32_t>* out, std::string* err);

// helpers on token streams
bool foam_tok_is_number(const std::string& t, double* v);
// reads the numbers of a (possibly counted, possibly nested one level: List<vector>) list starting at token index i ("(" or "N" "(")
// into out; ncomp = 1 for scalars, 3 for `( (x y z) (x y z) ... )`.  Returns false on malformed input.
bool foam_read_list(const std::vector<std::string>& tok, size_t i, int ncomp, std::vector<double>* out);

}  // namespace fy
