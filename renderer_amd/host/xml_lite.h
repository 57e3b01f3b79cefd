// xml_lite.h — the small subset of XML the renderer's scene / config / material files use:
// elements, attributes (Name = "..."), text, comments, several top-level elements per file.
// The accessor names follow the rapidxml calls the reference's ConfigManager makes
// (first_node / next_sibling / first_attribute / value, ConfigManager.cpp:166-433) so the loader
// reads like the reference's; the parser itself is new.
#pragma once
#include <cctype>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace mirt_host {

class XmlNode {
public:
    std::string name, text;
    std::vector<std::pair<std::string, std::string>> attrs;
    std::vector<std::unique_ptr<XmlNode>> children;
    XmlNode* parent = nullptr;
    size_t index_in_parent = 0;

    const char* value() const { return text.c_str(); }
    XmlNode* first_node(const char* tag = nullptr) const {
        for (auto& c : children)
            if (!tag || c->name == tag) return c.get();
        return nullptr;
    }
    XmlNode* next_sibling(const char* tag = nullptr) const {
        if (!parent) return nullptr;
        for (size_t i = index_in_parent + 1; i < parent->children.size(); i++)
            if (!tag || parent->children[i]->name == tag) return parent->children[i].get();
        return nullptr;
    }
    // nullptr when absent
    const std::string* first_attribute(const char* key) const {
        for (auto& a : attrs)
            if (a.first == key) return &a.second;
        return nullptr;
    }
};

// A document is an unnamed root whose children are the file's top-level elements.
class XmlDocument : public XmlNode {
public:
    void parse(const std::string& s) {
        children.clear();
        size_t p = 0;
        parse_children(this, s, p, true);
    }

private:
    static void skip_ws(const std::string& s, size_t& p) {
        while (p < s.size() && std::isspace((unsigned char)s[p])) p++;
    }
    static bool starts(const std::string& s, size_t p, const char* lit) {
        return s.compare(p, std::char_traits<char>::length(lit), lit) == 0;
    }
    static std::string trim(const std::string& t) {
        size_t a = 0, b = t.size();
        while (a < b && std::isspace((unsigned char)t[a])) a++;
        while (b > a && std::isspace((unsigned char)t[b - 1])) b--;
        return t.substr(a, b - a);
    }
    static std::string read_name(const std::string& s, size_t& p) {
        size_t b = p;
        while (p < s.size() && (std::isalnum((unsigned char)s[p]) || s[p] == '_' || s[p] == '-' ||
                                s[p] == ':' || s[p] == '.'))
            p++;
        if (p == b) throw std::runtime_error("xml: expected a name at offset " + std::to_string(b));
        return s.substr(b, p - b);
    }
    static void parse_children(XmlNode* node, const std::string& s, size_t& p, bool top) {
        std::string text;
        while (p < s.size()) {
            if (s[p] != '<') {
                text.push_back(s[p++]);
                continue;
            }
            if (starts(s, p, "<!--")) {
                size_t e = s.find("-->", p + 4);
                if (e == std::string::npos) throw std::runtime_error("xml: unterminated comment");
                p = e + 3;
                continue;
            }
            if (starts(s, p, "<?")) {
                size_t e = s.find("?>", p + 2);
                if (e == std::string::npos) throw std::runtime_error("xml: unterminated declaration");
                p = e + 2;
                continue;
            }
            if (starts(s, p, "</")) {
                if (top) throw std::runtime_error("xml: stray closing tag");
                p += 2;
                std::string nm = read_name(s, p);
                skip_ws(s, p);
                if (p >= s.size() || s[p] != '>' || nm != node->name)
                    throw std::runtime_error("xml: mismatched </" + nm + "> for <" + node->name + ">");
                p++;
                node->text = trim(text);
                return;
            }
            // element
            p++;
            std::unique_ptr<XmlNode> child(new XmlNode);
            child->name = read_name(s, p);
            child->parent = node;
            child->index_in_parent = node->children.size();
            bool self_closed = false;
            for (;;) {
                skip_ws(s, p);
                if (p >= s.size()) throw std::runtime_error("xml: unterminated tag <" + child->name);
                if (s[p] == '>') {
                    p++;
                    break;
                }
                if (starts(s, p, "/>")) {
                    p += 2;
                    self_closed = true;
                    break;
                }
                std::string key = read_name(s, p);
                skip_ws(s, p);
                if (p >= s.size() || s[p] != '=') throw std::runtime_error("xml: attribute without '='");
                p++;
                skip_ws(s, p);
                if (p >= s.size() || (s[p] != '"' && s[p] != '\''))
                    throw std::runtime_error("xml: attribute value must be quoted");
                char q = s[p++];
                size_t e = s.find(q, p);
                if (e == std::string::npos) throw std::runtime_error("xml: unterminated attribute value");
                child->attrs.emplace_back(key, s.substr(p, e - p));
                p = e + 1;
            }
            XmlNode* raw = child.get();
            node->children.push_back(std::move(child));
            if (!self_closed) parse_children(raw, s, p, false);
        }
        if (!top) throw std::runtime_error("xml: missing </" + node->name + ">");
    }
};

}  // namespace mirt_host
