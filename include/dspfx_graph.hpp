// dspfx_graph.hpp -- a saved graph of the reference (its `DSPConfig` JSON) on the engine, from C++.
//
// The reference GUI saves the graph as (dsp-stuff/src/runtime.rs:44-48, 560-564, 606-612)
//     {"nodes": [{"id": N, "typename": cfg_name, "position": [x, y], "cfg": {...}}],
//      "links": [{"lhs": [node_id, out_port_id], "rhs": [node_id, in_port_id]}]}
// where a node's `cfg` holds its id, its port maps {"inputs": {name: port_id}, "outputs": {...}} and every
// field tagged `save` (dsp-stuff-derive/src/lib.rs:233-293).  `SavedGraph` parses such a document, orders the nodes
// topologically (document order among the ready ones, like the Python mirror dsp-stuff_amd/graph.py) and hands the
// graph to `Engine::set_graph` (dspfx.h: dspfx_graph_set -- one generated kernel for the whole DAG).
//
// Same rules as the Python mirror: Mux / Demux are routing (identity on the selected port; a demux's unselected output
// is a connected pipe of zeros), the display-only nodes (pitch, wave_view, spectrogram) are dropped, muff is outside
// the accelerated path, a slider port takes at most one link, LowPass's cfg_name quirk (nodes/low_pass.rs:9: it saves
// itself as "high_pass") is the document's business.  Graphs that one kernel cannot hold (more than
// DSPFX_GRAPH_MAX_NODES nodes, a FIR or Fuzz node) are cut into a series of engines by `segment_plan`, the same cutting
// as graph.py's; what neither can cut is evaluated run by run (graph.py only).
#ifndef DSPFX_GRAPH_HPP
#define DSPFX_GRAPH_HPP

#include <algorithm>
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <string>
#include <utility>
#include <vector>

#include "dspfx.hpp"

namespace dspfx {

struct ConfigError : std::runtime_error {
    explicit ConfigError(const std::string &m) : std::runtime_error("DSPConfig: " + m) {}
};

// ---- the little JSON this needs ------------------------------------------------------------------------------------
namespace json {
struct Value {
    enum Type { Null, Bool, Number, String, Array, Object } type = Null;
    bool b = false;
    double num = 0.0;
    std::string str;
    std::vector<Value> arr;
    std::vector<std::pair<std::string, Value>> obj;   // document order
    const Value *find(const std::string &key) const {
        for (const auto &kv : obj)
            if (kv.first == key) return &kv.second;
        return nullptr;
    }
    const Value &at(const std::string &key) const {
        const Value *v = find(key);
        if (!v) throw ConfigError("missing field \"" + key + "\"");
        return *v;
    }
};
class Parser {
  public:
    explicit Parser(const std::string &s) : s_(s) {}
    Value parse() {
        Value v = value();
        ws();
        if (i_ != s_.size()) fail("trailing characters");
        return v;
    }

  private:
    const std::string &s_;
    std::size_t i_ = 0;
    [[noreturn]] void fail(const char *what) const { throw ConfigError(std::string("not JSON: ") + what + " at offset " + std::to_string(i_)); }
    void ws() {
        while (i_ < s_.size() && (s_[i_] == ' ' || s_[i_] == '\n' || s_[i_] == '\t' || s_[i_] == '\r')) ++i_;
    }
    bool lit(const char *w) {
        const std::size_t n = std::char_traits<char>::length(w);
        if (s_.compare(i_, n, w) != 0) return false;
        i_ += n;
        return true;
    }
    std::string string() {
        std::string out;
        ++i_;   // opening quote
        while (true) {
            if (i_ >= s_.size()) fail("unterminated string");
            const char c = s_[i_++];
            if (c == '"') return out;
            if (c != '\\') { out += c; continue; }
            if (i_ >= s_.size()) fail("unterminated escape");
            const char e = s_[i_++];
            switch (e) {
            case '"': case '\\': case '/': out += e; break;
            case 'b': out += '\b'; break;
            case 'f': out += '\f'; break;
            case 'n': out += '\n'; break;
            case 'r': out += '\r'; break;
            case 't': out += '\t'; break;
            case 'u': {   // names in these documents are ASCII; anything else is kept as UTF-8 of the code unit
                if (i_ + 4 > s_.size()) fail("short \\u escape");
                const unsigned cp = (unsigned)std::strtoul(s_.substr(i_, 4).c_str(), nullptr, 16);
                i_ += 4;
                if (cp < 0x80) out += (char)cp;
                else if (cp < 0x800) { out += (char)(0xC0 | (cp >> 6)); out += (char)(0x80 | (cp & 0x3F)); }
                else { out += (char)(0xE0 | (cp >> 12)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
                break;
            }
            default: fail("unknown escape");
            }
        }
    }
    Value value() {
        ws();
        if (i_ >= s_.size()) fail("unexpected end");
        Value v;
        const char c = s_[i_];
        if (c == '{') {
            v.type = Value::Object;
            ++i_;
            ws();
            if (i_ < s_.size() && s_[i_] == '}') { ++i_; return v; }
            while (true) {
                ws();
                if (i_ >= s_.size() || s_[i_] != '"') fail("expected a key");
                std::string k = string();
                ws();
                if (i_ >= s_.size() || s_[i_++] != ':') fail("expected ':'");
                v.obj.emplace_back(std::move(k), value());
                ws();
                if (i_ < s_.size() && s_[i_] == ',') { ++i_; continue; }
                if (i_ < s_.size() && s_[i_] == '}') { ++i_; return v; }
                fail("expected ',' or '}'");
            }
        }
        if (c == '[') {
            v.type = Value::Array;
            ++i_;
            ws();
            if (i_ < s_.size() && s_[i_] == ']') { ++i_; return v; }
            while (true) {
                v.arr.push_back(value());
                ws();
                if (i_ < s_.size() && s_[i_] == ',') { ++i_; continue; }
                if (i_ < s_.size() && s_[i_] == ']') { ++i_; return v; }
                fail("expected ',' or ']'");
            }
        }
        if (c == '"') { v.type = Value::String; v.str = string(); return v; }
        if (lit("true")) { v.type = Value::Bool; v.b = true; return v; }
        if (lit("false")) { v.type = Value::Bool; return v; }
        if (lit("null")) return v;
        char *end = nullptr;
        v.num = std::strtod(s_.c_str() + i_, &end);
        if (end == s_.c_str() + i_) fail("unexpected character");
        i_ = (std::size_t)(end - s_.c_str());
        v.type = Value::Number;
        return v;
    }
};
inline Value parse(const std::string &s) { return Parser(s).parse(); }
}  // namespace json

// ---- the saved graph -----------------------------------------------------------------------------------------------
class SavedGraph {
  public:
    static constexpr int ZERO = -1;   // pseudo producer: the unselected output of a demux (a connected pipe of zeros)
    struct GNode {
        int id = 0;
        std::string typename_;
        json::Value cfg;
        bool has_spec = false;
        Node spec;
        std::vector<int> main, side;                       // producer ids in link order
        std::map<int, std::vector<int>> ctl;               // slider index -> producers
        std::vector<std::pair<int, std::string>> outs;     // (consumer id, port name)
    };

    explicit SavedGraph(const std::string &text, bool page_round = false) {
        const json::Value doc = json::parse(text);
        if (doc.type != json::Value::Object) throw ConfigError("not a DSPConfig document");
        for (const json::Value &n : doc.at("nodes").arr) {
            GNode g;
            g.id = (int)n.at("id").num;
            g.typename_ = n.at("typename").str;
            g.cfg = n.at("cfg");
            const std::string &tn = g.typename_;
            if (tn == "pitch" || tn == "wave_view" || tn == "spectrogram") { dropped_.push_back(g.id); continue; }
            if (tn == "muff") throw ConfigError("node type \"muff\" is outside the accelerated path");
            if (tn == "mux" || tn == "demux") {
                const json::Value *sel = g.cfg.find(tn == "mux" ? "in_port" : "out_port");
                const std::string s = sel ? sel->str : "A";
                if (s != "A" && s != "B") throw ConfigError(tn + " node: unknown port selection " + s);
                g.has_spec = true;
                g.spec = Gain(1.0f);           // x * 1.0f: the copy_from_slice of mux.rs:54 / demux.rs:51,54
            } else if (tn != "input" && tn != "output") {
                g.has_spec = true;
                g.spec = spec_from_cfg(tn, g.cfg, page_round);
            }
            index_[g.id] = nodes_.size();
            nodes_.push_back(std::move(g));
        }
        for (const GNode &n : nodes_) {
            if (n.typename_ == "input") inputs_.push_back(n.id);
            if (n.typename_ == "output") outputs_.push_back(n.id);
        }
        if (inputs_.size() > 1 || outputs_.size() != 1) throw ConfigError("expected at most one input node and exactly one output node");
        for (const json::Value &l : doc.at("links").arr) add_link(l);
        for (const GNode &n : nodes_)
            for (const auto &kv : n.ctl)
                if (kv.second.size() > 1) throw ConfigError("slider port " + std::to_string(kv.first) + " of node " + std::to_string(n.id) + " averages several links");
        toposort();
    }

    const std::vector<int> &order() const { return order_; }
    const GNode &node(int id) const { return nodes_[index_.at(id)]; }

    // The graph as dspfx_graph_set takes it (nodes in topological order, links in the order collect_and_average adds
    // them); false when it cannot be one kernel: more than DSPFX_GRAPH_MAX_NODES nodes, a FIR or Fuzz node.
    bool fused_plan(std::vector<Node> &specs, std::vector<dspfx_graph_link> &links) const {
        std::vector<int> ord;
        for (int id : order_)
            if (node(id).has_spec) ord.push_back(id);
        if (ord.size() > DSPFX_GRAPH_MAX_NODES) return false;
        std::map<int, int> idx;
        for (std::size_t i = 0; i < ord.size(); ++i) {
            const dspfx_node_desc &d = node(ord[i]).spec.d;
            if (d.kind == DSPFX_FIR || (d.kind == DSPFX_DISTORT && d.mode == DSPFX_DIST_FUZZ)) return false;
            idx[ord[i]] = (int)i;
        }
        auto src = [&](int s) {
            if (s == ZERO) return (int)DSPFX_GRAPH_ZERO;
            return node(s).typename_ == "input" ? (int)DSPFX_GRAPH_INPUT : idx.at(s);
        };
        specs.clear();
        links.clear();
        for (int id : ord) {
            const GNode &n = node(id);
            specs.push_back(n.spec);
            for (int s : n.main) links.push_back({src(s), idx.at(id), DSPFX_PORT_MAIN});
            for (int s : n.side) links.push_back({src(s), idx.at(id), DSPFX_PORT_SIDE});
            for (const auto &kv : n.ctl)
                for (int s : kv.second) links.push_back({src(s), idx.at(id), DSPFX_PORT_SLIDER + kv.first});
        }
        for (int s : node(outputs_[0]).main) links.push_back({src(s), (int)ord.size(), DSPFX_PORT_MAIN});
        return true;
    }

    // Install the graph on an engine (Error with DSPFX_ERR_UNSUPPORTED when it needs cutting: segment_plan).
    void install(Engine &e) const {
        std::vector<Node> specs;
        std::vector<dspfx_graph_link> links;
        if (!fused_plan(specs, links)) throw Error(DSPFX_ERR_UNSUPPORTED, "this graph does not fit one kernel (see segment_plan)");
        e.set_graph(specs, links);
    }

    // One engine of a graph that was cut into a series (segment_plan).
    struct Step {
        enum Kind { GraphKernel, NodeAveraged, NodeHop } kind = GraphKernel;
        std::vector<Node> specs;                 // GraphKernel: the segment's nodes; otherwise the one FIR / Fuzz node
        std::vector<dspfx_graph_link> links;     // GraphKernel only
        int in_ref = -1;                         // whose output block this step reads: a step index, -1 = the graph's Input block
        int in2_ref = NO_REF;                    // GraphKernel: the block read as DSPFX_GRAPH_INPUT2 (`side`), when it reads one
        static constexpr int NO_REF = -2;
        bool reads_second_block() const {
            for (const dspfx_graph_link &l : links)
                if (l.src == DSPFX_GRAPH_INPUT2) return true;
            return false;
        }
    };

    // A graph too large for one kernel, or with FIR / Fuzz nodes (kernels of their own), as a SERIES of engines -- the
    // same cutting as dsp-stuff_amd/graph.py segment_plan, which documents it:
    //   * in front of a FIR / Fuzz node that all the live signal goes into: the segment's Output is that node's averaged
    //     main port (NodeAveraged: the node's engine takes it as it is);
    //   * the same with ONE signal that also goes on beside the node (wet / dry): the segment hands it over RAW, the node's
    //     engine applies the hop (NodeHop: link_flags = DSPFX_LINK_INPUT) and the next segment reads the node as Input
    //     and that signal as DSPFX_GRAPH_INPUT2 (`side`);
    //   * where a stretch exceeds max_nodes, at a point that a single NEW signal crosses (RAW handover).
    // At any boundary one OLDER signal (the segment's own Input or second block: the dry signal of a wet / dry rig) may stay
    // alive beside the new one; the next segment reads it as DSPFX_GRAPH_INPUT2.
    // false when no such cutting exists (the graph is then evaluated run by run: graph.py shows how).
    bool segment_plan(std::vector<Step> &steps, int max_nodes = DSPFX_GRAPH_MAX_NODES) const {
        steps.clear();
        const int out_id = outputs_[0];
        const int in_id = inputs_.empty() ? NONE : inputs_[0];
        std::map<int, bool> fed;
        for (int id : order_) {
            bool f = node(id).typename_ == "input";
            for (int p : producers(node(id))) f = f || fed[p];
            fed[id] = f;
        }
        std::vector<int> nodes, order;
        for (int id : order_)
            if (node(id).has_spec) nodes.push_back(id);
        // sources fed by nothing are evaluated right before their first consumer
        std::map<int, bool> placed;
        auto place = [&](auto &&self, int id) -> void {
            if (placed[id]) return;
            for (int p : producers(node(id)))
                if (node(p).has_spec && !fed[p]) self(self, p);
            placed[id] = true;
            order.push_back(id);
        };
        for (int id : nodes)
            if (fed[id]) place(place, id);
        for (int id : nodes) place(place, id);
        std::map<int, int> pos;
        for (std::size_t i = 0; i < order.size(); ++i) pos[order[i]] = (int)i;
        pos[out_id] = (int)order.size();
        std::map<int, std::vector<int>> users;
        auto note_users = [&](int id) {
            for (int p : producers(node(id))) users[p].push_back(pos[id]);
        };
        for (int id : order) note_users(id);
        note_users(out_id);
        auto read_at_or_after = [&](int v, int p) {
            const auto it = users.find(v);
            if (it == users.end()) return false;
            for (int u : it->second)
                if (u >= p) return true;
            return false;
        };
        int start = 0, cur_in = in_id, cur_in2 = NONE, ref_in = -1, ref_in2 = Step::NO_REF;
        auto live_after = [&](int hi, int p) {          // of {cur_in, cur_in2} + order[start:hi]: still read at p or later
            std::vector<int> live;
            if (cur_in != NONE && read_at_or_after(cur_in, p)) live.push_back(cur_in);
            if (cur_in2 != NONE && read_at_or_after(cur_in2, p)) live.push_back(cur_in2);
            for (int k = start; k < hi; ++k)
                if (read_at_or_after(order[(std::size_t)k], p)) live.push_back(order[(std::size_t)k]);
            return live;
        };
        bool ok = true;
        auto emit = [&](int lo, int hi, const std::vector<int> &sink, bool raw) -> int {
            Step st;
            std::map<int, int> idx;
            for (int k = lo; k < hi; ++k) idx[order[(std::size_t)k]] = k - lo;
            auto src = [&](int sv) {
                if (sv == ZERO) return (int)DSPFX_GRAPH_ZERO;
                if (sv == cur_in) return (int)DSPFX_GRAPH_INPUT;
                if (sv == cur_in2) return (int)DSPFX_GRAPH_INPUT2;
                const auto it = idx.find(sv);
                if (it == idx.end()) { ok = false; return 0; }   // a signal from further back: no series form
                return it->second;
            };
            for (int k = lo; k < hi; ++k) {
                const GNode &n = node(order[(std::size_t)k]);
                st.specs.push_back(n.spec);
                for (int sv : n.main) st.links.push_back({src(sv), k - lo, DSPFX_PORT_MAIN});
                for (int sv : n.side) st.links.push_back({src(sv), k - lo, DSPFX_PORT_SIDE});
                for (const auto &kv : n.ctl)
                    for (int sv : kv.second) st.links.push_back({src(sv), k - lo, DSPFX_PORT_SLIDER + kv.first});
            }
            for (int sv : sink) st.links.push_back({src(sv), hi - lo, DSPFX_PORT_MAIN | (raw ? DSPFX_PORT_RAW : 0)});
            st.in_ref = ref_in;
            st.in2_ref = st.reads_second_block() ? ref_in2 : Step::NO_REF;
            steps.push_back(std::move(st));
            return (int)steps.size() - 1;
        };
        auto ref_of = [&](int sv) { return sv == cur_in ? ref_in : ref_in2; };
        auto is_old = [&](int v) { return v == cur_in || v == cur_in2; };
        const int n = (int)order.size();
        for (int i = 0; i <= n && ok; ++i) {
            const bool at_end = i == n;
            const bool cut_node = !at_end && unfusable(node(order[(std::size_t)i]).spec);
            if (!at_end && !cut_node) continue;
            while (i - start > max_nodes) {            // cut the stretch where ONE new signal crosses while it does not fit
                int best_p = -1, best_v = NONE, best_old = NONE;
                for (int p = start + 1; p <= std::min(start + max_nodes, i - 1); ++p) {
                    std::vector<int> olds, news;
                    for (int v : live_after(p, p)) (is_old(v) ? olds : news).push_back(v);
                    if (news.size() == 1 && olds.size() <= 1) { best_p = p; best_v = news[0]; best_old = olds.empty() ? NONE : olds[0]; }
                }
                if (best_p < 0) return false;
                const int old_ref = best_old == NONE ? Step::NO_REF : ref_of(best_old);
                const int k = emit(start, best_p, {best_v}, true);
                cur_in2 = best_old;
                ref_in2 = old_ref;
                start = best_p;
                cur_in = best_v;
                ref_in = k;
            }
            if (at_end) {
                emit(start, i, node(out_id).main, false);
                break;
            }
            const GNode &u = node(order[(std::size_t)i]);
            if (!u.ctl.empty() || !u.side.empty()) return false;
            const std::vector<int> live = live_after(i, i);
            auto in_main = [&](int v) { return std::find(u.main.begin(), u.main.end(), v) != u.main.end(); };
            std::vector<int> carried, rest;            // carried: older signals going around the node
            for (int v : live) ((is_old(v) && !in_main(v)) ? carried : rest).push_back(v);
            bool all_into_u = carried.size() <= 1;
            for (int v : rest) all_into_u = all_into_u && in_main(v) && !read_at_or_after(v, i + 1);
            Step nd;
            nd.specs = {u.spec};
            if (all_into_u) {
                const int keep = carried.empty() ? NONE : carried[0], keep_ref = carried.empty() ? Step::NO_REF : ref_of(carried[0]);
                nd.in_ref = emit(start, i, u.main, false);
                nd.kind = Step::NodeAveraged;
                cur_in2 = keep;
                ref_in2 = keep_ref;
            } else if (live.size() == 1 && u.main == live &&
                       ((pos.count(live[0]) && pos[live[0]] >= start && pos[live[0]] < i && live[0] != out_id) || (live[0] == cur_in && start == i))) {
                nd.in_ref = start < i ? emit(start, i, live, true) : ref_in;
                nd.kind = Step::NodeHop;
                cur_in2 = live[0];
                ref_in2 = nd.in_ref;
            } else {
                return false;
            }
            steps.push_back(std::move(nd));
            start = i + 1;
            cur_in = order[(std::size_t)i];
            ref_in = (int)steps.size() - 1;
        }
        return ok;
    }

    // One step of the general plan (region_plan): a generated kernel over a REGION of the graph with several input and
    // output blocks, a FIR / Fuzz node between regions, or the Output node's average when no region is left to carry it.
    struct Ref {                                   // whose block: step >= 0 -> that step's output block `block`;
        int step = -1, block = 0;                  // step == -1 -> the graph's Input block; step == -2 -> a pipe of zeros
        bool operator==(const Ref &o) const { return step == o.step && block == o.block; }
    };
    struct RegionStep {
        enum Kind { Region, NodeStep, OutputAvg } kind = Region;
        std::vector<Node> specs;                   // Region: its nodes; NodeStep: the one node
        std::vector<dspfx_graph_link> links;       // Region: sources DSPFX_GRAPH_INPUT_N(k) = in_refs[k]; dst == specs.size() + m = output block m
        std::vector<Ref> in_refs;                  // Region
        int n_out = 1;                             // Region
        std::vector<Ref> main_refs;                // NodeStep: main port averaged over these; OutputAvg: the Output node's links
        std::map<int, Ref> ctl_refs;               // NodeStep: slider k <- block
    };

    // ANY graph as a short series of generated kernels -- the same plan as dsp-stuff_amd/graph.py region_plan, which
    // documents it: the evaluation order is cut into regions of at most max_nodes fusable nodes, each one kernel reading up
    // to max_io blocks (from the Input node, earlier regions, FIR / Fuzz nodes; as main inputs, Add / Mix side inputs or
    // control signals alike) and writing up to max_io blocks (every signal a later step still reads, handed over RAW); the
    // last region also evaluates the Output node's port.  false when some region would need more blocks either way.
    bool region_plan(std::vector<RegionStep> &steps, int max_nodes = DSPFX_GRAPH_MAX_NODES, int max_io = DSPFX_GRAPH_MAX_IO) const {
        steps.clear();
        const int out_id = outputs_[0];
        std::map<int, bool> fed;
        for (int id : order_) {
            bool f = node(id).typename_ == "input";
            for (int p : producers(node(id))) f = f || fed[p];
            fed[id] = f;
        }
        std::vector<int> nodes, order;
        for (int id : order_)
            if (node(id).has_spec) nodes.push_back(id);
        std::map<int, bool> placed;
        auto place = [&](auto &&self, int id) -> void {      // sources fed by nothing come right before their first consumer
            if (placed[id]) return;
            for (int p : producers(node(id)))
                if (node(p).has_spec && !fed[p]) self(self, p);
            placed[id] = true;
            order.push_back(id);
        };
        for (int id : nodes)
            if (fed[id]) place(place, id);
        for (int id : nodes) place(place, id);
        const int n = (int)order.size();
        std::map<int, int> pos, last_use;
        for (int i = 0; i < n; ++i) pos[order[(std::size_t)i]] = i;
        pos[out_id] = n;
        auto note = [&](int id) {
            for (int p : producers(node(id))) {
                const auto it = last_use.find(p);
                if (it == last_use.end() || it->second < pos[id]) last_use[p] = pos[id];
            }
        };
        for (int id : order) note(id);
        note(out_id);
        std::map<int, Ref> loc;
        if (!inputs_.empty()) loc[inputs_[0]] = Ref{-1, 0};
        auto ref = [&](int v) { return v == ZERO ? Ref{-2, 0} : loc.at(v); };
        int i = 0;
        bool carries_output = false;
        while (i < n) {
            const GNode &nd = node(order[(std::size_t)i]);
            if (unfusable(nd.spec)) {
                RegionStep st;
                st.kind = RegionStep::NodeStep;
                st.specs = {nd.spec};
                for (int v : nd.main) st.main_refs.push_back(ref(v));
                for (const auto &kv : nd.ctl) st.ctl_refs[kv.first] = ref(kv.second[0]);
                steps.push_back(std::move(st));
                loc[nd.id] = Ref{(int)steps.size() - 1, 0};
                carries_output = false;
                ++i;
                continue;
            }
            int best_end = -1;
            std::vector<int> best_ext, best_outs;
            bool best_final = false;
            for (int end = i + 1; end <= n && end - i <= max_nodes && !unfusable(node(order[(std::size_t)end - 1]).spec); ++end) {
                const bool final = end == n;
                std::vector<int> ext, outs;
                auto inside = [&](int v) { const auto it = pos.find(v); return it != pos.end() && it->second >= i && it->second < end && v != out_id; };
                auto scan = [&](int id) {
                    for (int v : producers(node(id)))
                        if (!inside(v) && std::find(ext.begin(), ext.end(), v) == ext.end()) ext.push_back(v);
                };
                for (int k = i; k < end; ++k) scan(order[(std::size_t)k]);
                if (final) scan(out_id);
                if (!final)
                    for (int k = i; k < end; ++k) {
                        const auto it = last_use.find(order[(std::size_t)k]);
                        if (it != last_use.end() && it->second >= end) outs.push_back(order[(std::size_t)k]);
                    }
                const int n_out = (final ? 1 : 0) + (int)outs.size();
                if ((int)ext.size() <= max_io && n_out <= max_io) {
                    best_end = end;
                    best_ext = ext;
                    best_outs = outs;
                    best_final = final;
                }
            }
            if (best_end < 0) return false;
            RegionStep st;
            std::map<int, int> idx;
            for (int k = i; k < best_end; ++k) idx[order[(std::size_t)k]] = k - i;
            auto src = [&](int v) {
                if (v == ZERO) return (int)DSPFX_GRAPH_ZERO;
                const auto it = idx.find(v);
                if (it != idx.end()) return it->second;
                const int blk = (int)(std::find(best_ext.begin(), best_ext.end(), v) - best_ext.begin());
                return (int)DSPFX_GRAPH_INPUT_N(blk);
            };
            for (int k = i; k < best_end; ++k) {
                const GNode &m = node(order[(std::size_t)k]);
                st.specs.push_back(m.spec);
                for (int v : m.main) st.links.push_back({src(v), k - i, DSPFX_PORT_MAIN});
                for (int v : m.side) st.links.push_back({src(v), k - i, DSPFX_PORT_SIDE});
                for (const auto &kv : m.ctl)
                    for (int v : kv.second) st.links.push_back({src(v), k - i, DSPFX_PORT_SLIDER + kv.first});
            }
            const int nn = best_end - i;
            if (best_final)
                for (int v : node(out_id).main) st.links.push_back({src(v), nn, DSPFX_PORT_MAIN});
            if (!best_final && best_outs.empty()) best_outs.push_back(order[(std::size_t)best_end - 1]);   // a dead branch: still one block
            const int base = best_final ? 1 : 0;
            for (std::size_t m = 0; m < best_outs.size(); ++m)
                st.links.push_back({idx.at(best_outs[m]), nn + base + (int)m, DSPFX_PORT_MAIN | DSPFX_PORT_RAW});
            for (int v : best_ext) st.in_refs.push_back(loc.at(v));
            st.n_out = base + (int)best_outs.size();
            steps.push_back(std::move(st));
            for (std::size_t m = 0; m < best_outs.size(); ++m) loc[best_outs[m]] = Ref{(int)steps.size() - 1, base + (int)m};
            carries_output = best_final;
            i = best_end;
        }
        if (!carries_output) {
            RegionStep st;
            st.kind = RegionStep::OutputAvg;
            for (int v : node(out_id).main) st.main_refs.push_back(ref(v));
            steps.push_back(std::move(st));
        }
        return true;
    }

  private:
    static constexpr int NONE = -1000000;      // "no such signal" (node ids are the document's, ZERO is -1)
    static bool unfusable(const Node &n) { return n.d.kind == DSPFX_FIR || (n.d.kind == DSPFX_DISTORT && n.d.mode == DSPFX_DIST_FUZZ); }
    struct Row {
        int kind;
        std::vector<std::string> fields;
        std::string main_port;                  // "" = a source without a main port
        std::vector<std::string> ctl_ports;
    };
    // typename -> (kind, saved slider fields in params order, main input port, `as_input` control ports)
    static const std::map<std::string, Row> &table() {
        static const std::map<std::string, Row> t = {
            {"gain", {DSPFX_GAIN, {"level"}, "in", {"level"}}},
            {"biquad", {DSPFX_BIQUAD, {"a0", "a1", "a2", "b0", "b1", "b2"}, "in", {}}},
            {"low_pass", {DSPFX_LOW_PASS, {"ratio"}, "in", {}}},
            {"high_pass", {DSPFX_HIGH_PASS, {"ratio"}, "in", {}}},
            {"reverb", {DSPFX_REVERB, {"decay"}, "in", {}}},
            {"distort", {DSPFX_DISTORT, {"level"}, "in", {"level"}}},
            {"overdrive", {DSPFX_OVERDRIVE, {"boost", "drive", "level"}, "in", {"boost", "drive", "level"}}},
            {"chebyshev", {DSPFX_CHEBYSHEV, {"level_pos", "level_neg"}, "in", {}}},
            {"fir", {DSPFX_FIR, {}, "in", {}}},
            {"add", {DSPFX_ADD, {}, "a", {}}},
            {"mix", {DSPFX_MIX, {"ratio"}, "a", {"ratio"}}},
            {"envelope", {DSPFX_ENVELOPE, {"attack", "release"}, "in", {}}},
            {"signal_gen", {DSPFX_SIGNAL_GEN, {"amplitude", "frequency"}, "", {"amplitude", "frequency"}}},
        };
        return t;
    }
    static int index_of(const std::vector<std::string> &names, const std::string &what, const char *kind) {
        const auto it = std::find(names.begin(), names.end(), what);
        if (it == names.end()) throw ConfigError(std::string("unknown ") + kind + " \"" + what + "\"");
        return (int)(it - names.begin());
    }
    static Node spec_from_cfg(const std::string &tn, const json::Value &cfg, bool page_round) {
        const auto it = table().find(tn);
        if (it == table().end()) throw ConfigError("unknown node type \"" + tn + "\"");
        const Row &row = it->second;
        Node n = make(row.kind);
        for (std::size_t k = 0; k < row.fields.size(); ++k) {
            const json::Value *v = cfg.find(row.fields[k]);
            if (!v) throw ConfigError(tn + " node lacks saved field \"" + row.fields[k] + "\"");
            n.d.params[k] = (float)v->num;
        }
        if (row.kind == DSPFX_REVERB) {   // restoring a reverb runs refresh_seconds (lib.rs:319-337): reverb.rs:58
            n.d.params[1] = (float)cfg.at("seconds").num;      // the slider travels with the node (a later store refreshes from it)
            n.d.mode = page_round ? 1 : 0;
            n.d.delay_len = dspfx_delay_len((float)cfg.at("seconds").num, page_round ? 1 : 0);
        }
        if (row.kind == DSPFX_DISTORT) {
            const json::Value *m = cfg.find("mode");
            n.d.mode = index_of({"HardClip", "SoftClip", "Tanh", "RecipSoftClip", "Fuzz", "Sin", "Atan", "Square", "Chebyshev4"},
                                m ? m->str : "SoftClip", "distort mode");
        }
        if (row.kind == DSPFX_SIGNAL_GEN) {
            const json::Value *m = cfg.find("mode");
            n.d.mode = index_of({"Sine", "Triangle", "Square", "Constant"}, m ? m->str : "Sine", "signal_gen mode");
        }
        if (row.kind == DSPFX_FIR) {
            const json::Value *m = cfg.find("mode");
            n.d.mode = index_of({"Balanced", "Average"}, m ? m->str : "Balanced", "fir mode");
            const json::Value *t = cfg.find("taps");          // stored time-reversed (fir.rs:163,168)
            if (!t) n.taps = {1.0};
            else
                for (const json::Value &x : t->arr) n.taps.push_back(x.num);
            if (n.taps.empty()) throw ConfigError("fir node needs a non-empty taps array");
        }
        return n;
    }
    GNode &mut(int id) {
        const auto it = index_.find(id);
        if (it == index_.end()) throw ConfigError("link refers to a missing node");
        return nodes_[it->second];
    }
    static std::string port_name(const GNode &n, int port_id, const char *which) {
        const json::Value *m = n.cfg.find(which);
        if (m)
            for (const auto &kv : m->obj)
                if ((int)kv.second.num == port_id) return kv.first;
        throw ConfigError(std::string("link refers to an unknown port ") + std::to_string(port_id) + " of node " + std::to_string(n.id));
    }
    static std::string lower(std::string s) {
        for (char &c : s) c = (char)std::tolower((unsigned char)c);
        return s;
    }
    static std::string sel(const GNode &n, const char *field) {
        const json::Value *v = n.cfg.find(field);
        return lower(v ? v->str : "A");
    }
    void add_link(const json::Value &l) {
        int ln = (int)l.at("lhs").arr.at(0).num;
        const int lp = (int)l.at("lhs").arr.at(1).num, rn = (int)l.at("rhs").arr.at(0).num, rp = (int)l.at("rhs").arr.at(1).num;
        if (std::find(dropped_.begin(), dropped_.end(), rn) != dropped_.end()) return;
        GNode &src = mut(ln);
        GNode &dst = mut(rn);
        const std::string oname = port_name(src, lp, "outputs"), pname = port_name(dst, rp, "inputs");
        if (src.typename_ == "demux" && oname != sel(src, "out_port")) ln = ZERO;   // demux.rs:49-56: the other output stays zeroed
        else src.outs.emplace_back(rn, pname);
        if (dst.typename_ == "output") { dst.main.push_back(ln); return; }
        if (dst.typename_ == "mux") {
            if (pname == sel(dst, "in_port")) dst.main.push_back(ln);
            else if (ln != ZERO) src.outs.back().second = "unused";   // mux.rs:47-50: the unselected port is never read
            return;
        }
        if (dst.typename_ == "demux") { dst.main.push_back(ln); return; }
        const Row &row = table().at(dst.typename_);
        const auto ctl = std::find(row.ctl_ports.begin(), row.ctl_ports.end(), pname);
        if (!row.main_port.empty() && pname == row.main_port) dst.main.push_back(ln);
        else if (pname == "b" && (row.kind == DSPFX_ADD || row.kind == DSPFX_MIX)) dst.side.push_back(ln);
        else if (ctl != row.ctl_ports.end()) dst.ctl[index_of(row.fields, pname, "slider")].push_back(ln);
        else throw ConfigError("node " + std::to_string(rn) + " (" + dst.typename_ + ") has no input port \"" + pname + "\"");
    }
    std::vector<int> producers(const GNode &n) const {
        std::vector<int> p;
        for (int s : n.main) if (s != ZERO) p.push_back(s);
        for (int s : n.side) if (s != ZERO) p.push_back(s);
        for (const auto &kv : n.ctl)
            for (int s : kv.second) if (s != ZERO) p.push_back(s);
        return p;
    }
    void toposort() {
        std::map<int, int> indeg;
        std::vector<int> ready;
        for (const GNode &n : nodes_) {
            indeg[n.id] = (int)producers(n).size();
            if (indeg[n.id] == 0) ready.push_back(n.id);          // document order: deterministic
        }
        for (std::size_t k = 0; k < ready.size(); ++k) {
            const int i = ready[k];
            order_.push_back(i);
            for (const auto &o : node(i).outs) {
                if (o.second == "unused") continue;
                if (--indeg[o.first] == 0) ready.push_back(o.first);
            }
        }
        if (order_.size() != nodes_.size()) throw ConfigError("graph has a cycle");
    }

    std::vector<GNode> nodes_;
    std::map<int, std::size_t> index_;
    std::vector<int> dropped_, inputs_, outputs_, order_;
};

// Write a chain in the reference's format (input -> chain -> output, fresh ids), the way File > Save would
// (runtime.rs:466-479, 606-612) -- the counterpart of dsp-stuff_amd/config.py dump_dspconfig.  A reverb's saved slider is
// delay_len / 48000 s.  With faithful_lowpass_bug a LowPass is written as the reference writes it: typename "high_pass"
// (nodes/low_pass.rs:9), which restores as a HighPass; false writes "low_pass".
inline std::string dump_dspconfig(const std::vector<Node> &chain, bool faithful_lowpass_bug = true) {
    static const char *typenames[] = {"gain", "biquad", "low_pass", "high_pass", "reverb", "distort", "overdrive", "chebyshev",
                                      "fir", "add", "mix", "signal_gen", "envelope"};
    static const std::vector<std::vector<std::string>> fields = {
        {"level"}, {"a0", "a1", "a2", "b0", "b1", "b2"}, {"ratio"}, {"ratio"}, {"decay"}, {"level"}, {"boost", "drive", "level"},
        {"level_pos", "level_neg"}, {}, {}, {"ratio"}, {"amplitude", "frequency"}, {"attack", "release"}};
    static const std::vector<std::vector<std::string>> ctl_ports = {
        {"level"}, {}, {}, {}, {}, {"level"}, {"boost", "drive", "level"}, {}, {}, {}, {"ratio"}, {"amplitude", "frequency"}, {}};
    static const char *distort_modes[] = {"HardClip", "SoftClip", "Tanh", "RecipSoftClip", "Fuzz", "Sin", "Atan", "Square", "Chebyshev4"};
    static const char *signal_modes[] = {"Sine", "Triangle", "Square", "Constant"};
    auto num = [](double v, int digits) {
        char b[64];
        std::snprintf(b, sizeof b, "%.*g", digits, v);
        std::string s(b);
        if (s.find_first_of(".eE") == std::string::npos && s.find("inf") == std::string::npos && s.find("nan") == std::string::npos) s += ".0";
        return s;
    };
    for (std::size_t k = 1; k < chain.size(); ++k)
        if (chain[k].d.kind == DSPFX_SIGNAL_GEN) throw ConfigError("a signal_gen node can only be the first node of a chain");
    const bool generator_first = !chain.empty() && chain[0].d.kind == DSPFX_SIGNAL_GEN;
    for (const Node &n : chain)
        if ((n.d.kind == DSPFX_ADD || n.d.kind == DSPFX_MIX) && generator_first)
            throw ConfigError("add/mix take their 'b' port from the input node, which a generator-sourced patch lacks");
    int next_id = 0;
    std::string nodes, links;
    int in_id = -1, in_port = -1, prev_node = -1, prev_port = -1;
    if (!generator_first) {
        in_id = next_id++;
        in_port = next_id++;
        nodes += "{\"id\": " + std::to_string(in_id) + ", \"typename\": \"input\", \"position\": [0.0, 0.0], \"cfg\": {\"id\": " +
                 std::to_string(in_id) + ", \"selected_host\": \"ALSA\", \"selected_device\": null, \"outputs\": {\"out\": " + std::to_string(in_port) + "}}}";
        prev_node = in_id;
        prev_port = in_port;
    }
    auto link = [&](int ln, int lp, int rn, int rp) {
        if (!links.empty()) links += ", ";
        links += "{\"lhs\": [" + std::to_string(ln) + ", " + std::to_string(lp) + "], \"rhs\": [" + std::to_string(rn) + ", " + std::to_string(rp) + "]}";
    };
    for (std::size_t k = 0; k < chain.size(); ++k) {
        const Node &n = chain[k];
        const int kind = n.d.kind;
        if (kind < 0 || kind >= DSPFX_N_KINDS) throw ConfigError("unknown node kind");
        std::string tn = typenames[kind];
        if (kind == DSPFX_LOW_PASS && faithful_lowpass_bug) tn = "high_pass";
        const std::string main_port = kind == DSPFX_SIGNAL_GEN ? "" : (kind == DSPFX_ADD || kind == DSPFX_MIX) ? "a" : "in";
        const int node_id = next_id++;
        std::string ins;
        int main_id = -1, b_id = -1;
        auto add_in = [&](const std::string &name, int id) {
            if (!ins.empty()) ins += ", ";
            ins += "\"" + name + "\": " + std::to_string(id);
        };
        if (!main_port.empty()) add_in(main_port, main_id = next_id++);
        if (kind == DSPFX_ADD || kind == DSPFX_MIX) add_in("b", b_id = next_id++);
        for (const std::string &cp : ctl_ports[(std::size_t)kind]) add_in(cp, next_id++);
        const int out_port = next_id++;
        std::string cfg = "\"id\": " + std::to_string(node_id) + ", \"inputs\": {" + ins + "}, \"outputs\": {\"out\": " + std::to_string(out_port) + "}";
        for (std::size_t f = 0; f < fields[(std::size_t)kind].size(); ++f) cfg += ", \"" + fields[(std::size_t)kind][f] + "\": " + num(n.d.params[f], 9);
        if (kind == DSPFX_REVERB) cfg += ", \"seconds\": " + num(n.d.params[1] > 0.0f ? n.d.params[1] : (float)(n.d.delay_len / 48000.0), 9);
        if (kind == DSPFX_DISTORT) cfg += std::string(", \"mode\": \"") + distort_modes[n.d.mode] + "\"";
        if (kind == DSPFX_SIGNAL_GEN) cfg += std::string(", \"mode\": \"") + signal_modes[n.d.mode] + "\"";
        if (kind == DSPFX_FIR) {
            cfg += std::string(", \"mode\": \"") + (n.d.mode == DSPFX_FIR_AVERAGE ? "Average" : "Balanced") + "\", \"file_name\": null, \"taps\": [";
            for (std::size_t t = 0; t < n.taps.size(); ++t) cfg += (t ? ", " : "") + num(n.taps[t], 17);
            cfg += "]";
        }
        if (!nodes.empty()) nodes += ", ";
        nodes += "{\"id\": " + std::to_string(node_id) + ", \"typename\": \"" + tn + "\", \"position\": [" + num(120.0 * (double)(k + 1), 9) +
                 ", 0.0], \"cfg\": {" + cfg + "}}";
        if (!main_port.empty()) link(prev_node, prev_port, node_id, main_id);
        if (kind == DSPFX_ADD || kind == DSPFX_MIX) link(in_id, in_port, node_id, b_id);
        prev_node = node_id;
        prev_port = out_port;
    }
    const int out_id = next_id++, out_port = next_id++;
    if (!nodes.empty()) nodes += ", ";
    nodes += "{\"id\": " + std::to_string(out_id) + ", \"typename\": \"output\", \"position\": [" + num(120.0 * (double)(chain.size() + 1), 9) +
             ", 0.0], \"cfg\": {\"id\": " + std::to_string(out_id) + ", \"selected_host\": \"ALSA\", \"selected_device\": null, \"inputs\": {\"in\": " + std::to_string(out_port) + "}}}";
    if (prev_node >= 0) link(prev_node, prev_port, out_id, out_port);
    return "{\"nodes\": [" + nodes + "], \"links\": [" + links + "]}";
}

}  // namespace dspfx

#endif  // DSPFX_GRAPH_HPP
