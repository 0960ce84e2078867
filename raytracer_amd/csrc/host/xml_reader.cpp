#include "xml_reader.h"

#include <cctype>
#include <cstring>

namespace rt {
namespace {

struct Cursor {
    const std::string &s;
    size_t i = 0;
    std::string err;

    bool eof() const { return i >= s.size(); }
    char peek() const { return eof() ? '\0' : s[i]; }
    bool starts(const char *lit) const { return s.compare(i, std::strlen(lit), lit) == 0; }
    void skip_ws()
    {
        while (!eof() && std::isspace((unsigned char)s[i])) ++i;
    }
    bool skip_until(const char *lit)
    {
        size_t p = s.find(lit, i);
        if (p == std::string::npos) return false;
        i = p + std::strlen(lit);
        return true;
    }
    bool fail(const std::string &m)
    {
        if (err.empty()) {
            size_t line = 1;
            for (size_t k = 0; k < i && k < s.size(); ++k) line += s[k] == '\n';
            err = m + " (line " + std::to_string(line) + ")";
        }
        return false;
    }
};

bool is_name_char(char c) { return std::isalnum((unsigned char)c) || c == '_' || c == '-' || c == ':' || c == '.'; }

std::string decode_entities(const std::string &v)
{
    std::string o;
    for (size_t i = 0; i < v.size(); ++i) {
        if (v[i] != '&') { o += v[i]; continue; }
        static const struct { const char *e; char c; } tab[] = {
            {"&amp;", '&'}, {"&lt;", '<'}, {"&gt;", '>'}, {"&quot;", '"'}, {"&apos;", '\''}};
        bool done = false;
        for (auto &t : tab)
            if (v.compare(i, std::strlen(t.e), t.e) == 0) { o += t.c; i += std::strlen(t.e) - 1; done = true; break; }
        if (!done) o += '&';
    }
    return o;
}

// skips comments, <? ?>, <! > and text; stops at '<' of an element or end tag
bool skip_misc(Cursor &c)
{
    for (;;) {
        while (!c.eof() && c.peek() != '<') ++c.i;
        if (c.eof()) return true;
        if (c.starts("<!--")) { if (!c.skip_until("-->")) return c.fail("unterminated comment"); }
        else if (c.starts("<?")) { if (!c.skip_until("?>")) return c.fail("unterminated processing instruction"); }
        else if (c.starts("<![CDATA[")) { if (!c.skip_until("]]>")) return c.fail("unterminated CDATA"); }
        else if (c.starts("<!")) { if (!c.skip_until(">")) return c.fail("unterminated declaration"); }
        else return true;
    }
}

bool parse_element(Cursor &c, XmlElement &parent, int depth)
{
    if (depth > 256) return c.fail("element nesting too deep");
    ++c.i; // '<'
    auto e = std::make_unique<XmlElement>();
    while (!c.eof() && is_name_char(c.peek())) e->tag += c.s[c.i++];
    if (e->tag.empty()) return c.fail("missing element name");
    for (;;) {
        c.skip_ws();
        if (c.eof()) return c.fail("unterminated start tag");
        if (c.starts("/>")) { c.i += 2; parent.children.push_back(std::move(e)); return true; }
        if (c.peek() == '>') { ++c.i; break; }
        std::string name;
        while (!c.eof() && is_name_char(c.peek())) name += c.s[c.i++];
        if (name.empty()) return c.fail("malformed attribute");
        c.skip_ws();
        if (c.peek() != '=') return c.fail("attribute without value");
        ++c.i;
        c.skip_ws();
        const char q = c.peek();
        if (q != '"' && q != '\'') return c.fail("attribute value must be quoted");
        ++c.i;
        size_t end = c.s.find(q, c.i);
        if (end == std::string::npos) return c.fail("unterminated attribute value");
        e->attrs.emplace_back(name, decode_entities(c.s.substr(c.i, end - c.i)));
        c.i = end + 1;
    }
    for (;;) {
        if (!skip_misc(c)) return false;
        if (c.eof()) return c.fail("missing </" + e->tag + ">");
        if (c.starts("</")) {
            c.i += 2;
            std::string name;
            while (!c.eof() && is_name_char(c.peek())) name += c.s[c.i++];
            c.skip_ws();
            if (c.peek() != '>') return c.fail("malformed end tag");
            ++c.i;
            if (name != e->tag) return c.fail("mismatched end tag </" + name + "> for <" + e->tag + ">");
            parent.children.push_back(std::move(e));
            return true;
        }
        if (!parse_element(c, *e, depth + 1)) return false;
    }
}

} // namespace

std::unique_ptr<XmlElement> xml_parse(const std::string &text, std::string &err)
{
    Cursor c{text};
    auto doc = std::make_unique<XmlElement>();
    for (;;) {
        if (!skip_misc(c)) { err = c.err; return nullptr; }
        if (c.eof()) break;
        if (c.starts("</")) { c.fail("unexpected end tag"); err = c.err; return nullptr; }
        if (!parse_element(c, *doc, 0)) { err = c.err; return nullptr; }
    }
    if (doc->children.empty()) { err = "document has no elements"; return nullptr; }
    return doc;
}

} // namespace rt
