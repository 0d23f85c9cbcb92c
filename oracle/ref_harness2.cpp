// ORACLE — TEST INFRASTRUCTURE ONLY.  C entry points over more of the REFERENCE's own sources, compiled in place from /root/reference (never
// copied): utils/vector_map.h, utils/sgf_loader.{h,cpp}, actor/search.h, environment/go/go_unit.h — the remaining units of the self-play /
// record path that build in this image without Boost.  (<stdexcept> is included first because vector_map.h uses std::out_of_range without
// including it; in the reference's own build another header brings it in.)  Output: oracle/_ref/libmzref.so (git-ignored).
#include <stdexcept>
#include "vector_map.h"
#include "sgf_loader.h"
#include "search.h"
#include "go_unit.h"
#include <cstring>
#include <sstream>
#include <string>

using namespace minizero;

static int copyOutRef(const std::string& s, char* buf, int cap)
{
    int n = static_cast<int>(s.size());
    if (buf && cap > 0) {
        int m = n < cap - 1 ? n : cap - 1;
        memcpy(buf, s.data(), m);
        buf[m] = 0;
    }
    return n;
}

extern "C" {

// same rendering as mzo_sgf_parse (oracle/o_capi.cpp)
int mzref_sgf_parse(const char* content, char* buf, int cap)
{
    utils::SGFLoader loader;
    const bool ok = loader.loadFromString(content);
    std::ostringstream o;
    o << (ok ? "ok" : "fail") << "|";
    for (auto& t : loader.getTags()) { o << t.first << "=" << t.second << ";"; }
    o << "|";
    for (auto& a : loader.getActions()) {
        o << a.first[0] << ":" << a.first[1] << "{";
        for (auto& t : a.second) { o << t.first << "=" << t.second << ";"; }
        o << "}";
    }
    return copyOutRef(o.str(), buf, cap);
}

// same as mzo_tagmap_apply
int mzref_tagmap_apply(const char* ops, char* buf, int cap)
{
    utils::VectorMap<std::string, std::string> m;
    std::istringstream iss(ops);
    std::string line;
    while (std::getline(iss, line)) {
        std::istringstream ls(line);
        std::string op, k, v;
        ls >> op >> k >> v;
        if (op == "set") { m[k] = v; }
        else if (op == "insert") { m.insert({k, v}); }
        else if (op == "erase") { m.erase(k); }
    }
    std::ostringstream o;
    for (auto& t : m) { o << t.first << "[" << t.second << "]"; }
    return copyOutRef(o.str(), buf, cap);
}

int mzref_sgf_coords(int, int board_size, const char* coord, const char* sgf, int* out)
{
    out[0] = utils::SGFLoader::boardCoordinateStringToActionID(coord, board_size);
    out[1] = utils::SGFLoader::sgfStringToActionID(sgf, board_size);
    return 0;
}
int mzref_sgf_strings(int action_id, int board_size, char* buf, int cap)
{
    return copyOutRef(utils::SGFLoader::actionIDToBoardCoordinateString(action_id, board_size) + "|" + utils::SGFLoader::actionIDToSGFString(action_id, board_size), buf, cap);
}
int mzref_go_constants(int* out)
{
    out[0] = env::go::kMaxGoBoardSize;
    out[1] = env::go::kGoNumPlayer;
    out[2] = static_cast<int>(sizeof(env::go::GoHashKey));
    out[3] = static_cast<int>(env::go::GoBitboard().size());
    struct S : actor::Search { void reset() override {} } s; // actor/search.h: the abstract base instantiates
    s.reset();
    return copyOutRef(env::go::kGoName, nullptr, 0);
}

} // extern "C"
