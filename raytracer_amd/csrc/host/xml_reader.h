// Minimal XML element-tree reader: elements, attributes, comments, processing
// instructions and character data (ignored).  Enough for the scene.xml schema
// the reference reads through tinyxml2 (src/xmlload.cpp, src/xmlload.h).
#pragma once
#include <memory>
#include <string>
#include <utility>
#include <vector>

namespace rt {

struct XmlElement {
    std::string tag;
    std::vector<std::pair<std::string, std::string>> attrs;
    std::vector<std::unique_ptr<XmlElement>> children;

    const char *attr(const char *name) const
    {
        for (auto &a : attrs)
            if (a.first == name) return a.second.c_str();
        return nullptr;
    }
    const XmlElement *child(const char *name) const
    {
        for (auto &c : children)
            if (c->tag == name) return c.get();
        return nullptr;
    }
};

// Parses a whole document; returns the root elements' container (tag empty).
// On failure returns nullptr and sets err.
std::unique_ptr<XmlElement> xml_parse(const std::string &text, std::string &err);

} // namespace rt
