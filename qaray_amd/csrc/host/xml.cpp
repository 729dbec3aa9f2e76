// xml.cpp — see xml.h.
#include "xml.h"

#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>

namespace qaray_hip {

const char *XmlElement::Attribute(const char *name) const
{
  for (const auto &a : attrs_) if (a.first == name) return a.second.c_str();
  return nullptr;
}

bool XmlElement::QueryDoubleAttribute(const char *name, double *out) const
{
  const char *v = Attribute(name);
  if (!v) return false;
  double d;
  if (sscanf(v, "%lf", &d) != 1) return false;
  *out = d;
  return true;
}

bool XmlElement::QueryIntAttribute(const char *name, int *out) const
{
  const char *v = Attribute(name);
  if (!v) return false;
  int d;
  if (sscanf(v, "%d", &d) != 1) return false;
  *out = d;
  return true;
}

const XmlElement *XmlElement::FirstChildElement(const char *name) const
{
  for (const auto &c : children_) if (c->name_ == name) return c.get();
  return nullptr;
}

const XmlElement *XmlDocument::FirstChildElement(const char *name) const
{
  for (const auto &c : roots_) if (c->Value() == name) return c.get();
  return nullptr;
}

bool XmlDocument::LoadFile(const char *filename)
{
  std::ifstream f(filename, std::ios::binary);
  if (!f) { error_ = std::string("cannot open ") + filename; return false; }
  std::stringstream ss;
  ss << f.rdbuf();
  return Parse(ss.str());
}

namespace {

struct Cursor {
  const std::string &s;
  size_t i = 0;
  explicit Cursor(const std::string &t) : s(t) {}
  bool Eof() const { return i >= s.size(); }
  bool StartsWith(const char *lit) const { return s.compare(i, strlen(lit), lit) == 0; }
  void SkipSpace() { while (!Eof() && isspace((unsigned char) s[i])) ++i; }
  bool SkipPast(const char *lit)
  {
    size_t p = s.find(lit, i);
    if (p == std::string::npos) return false;
    i = p + strlen(lit);
    return true;
  }
};

bool IsNameChar(char c) { return isalnum((unsigned char) c) || c == '_' || c == '-' || c == ':' || c == '.'; }

std::string DecodeEntities(const std::string &v)
{
  if (v.find('&') == std::string::npos) return v;
  static const struct { const char *ent; char ch; } table[] = {
      {"&amp;", '&'}, {"&lt;", '<'}, {"&gt;", '>'}, {"&quot;", '"'}, {"&apos;", '\''}};
  std::string out;
  for (size_t i = 0; i < v.size();) {
    bool hit = false;
    if (v[i] == '&') {
      for (const auto &e : table) {
        const size_t n = strlen(e.ent);
        if (v.compare(i, n, e.ent) == 0) { out.push_back(e.ch); i += n; hit = true; break; }
      }
    }
    if (!hit) out.push_back(v[i++]);
  }
  return out;
}

}  // namespace

bool XmlDocument::Parse(const std::string &text)
{
  roots_.clear();
  error_.clear();
  Cursor c(text);
  std::vector<XmlElement *> open;  // element stack
  auto fail = [&](const std::string &why) {
    error_ = why + " at byte " + std::to_string(c.i);
    roots_.clear();
    return false;
  };
  while (true) {
    // text between tags is ignored
    size_t lt = text.find('<', c.i);
    if (lt == std::string::npos) break;
    c.i = lt;
    if (c.StartsWith("<!--")) { if (!c.SkipPast("-->")) return fail("unterminated comment"); continue; }
    if (c.StartsWith("<?")) { if (!c.SkipPast("?>")) return fail("unterminated declaration"); continue; }
    if (c.StartsWith("<![CDATA[")) { if (!c.SkipPast("]]>")) return fail("unterminated CDATA"); continue; }
    if (c.StartsWith("<!")) { if (!c.SkipPast(">")) return fail("unterminated <!"); continue; }
    if (c.StartsWith("</")) {
      c.i += 2;
      size_t b = c.i;
      while (!c.Eof() && IsNameChar(text[c.i])) ++c.i;
      const std::string name = text.substr(b, c.i - b);
      c.SkipSpace();
      if (c.Eof() || text[c.i] != '>') return fail("malformed end tag");
      ++c.i;
      if (open.empty() || open.back()->name_ != name) return fail("mismatched end tag </" + name + ">");
      open.pop_back();
      continue;
    }
    // start tag
    ++c.i;
    size_t b = c.i;
    while (!c.Eof() && IsNameChar(text[c.i])) ++c.i;
    if (c.i == b) return fail("malformed start tag");
    std::unique_ptr<XmlElement> el(new XmlElement);
    el->name_ = text.substr(b, c.i - b);
    bool selfClosed = false;
    while (true) {
      c.SkipSpace();
      if (c.Eof()) return fail("unterminated start tag");
      if (text[c.i] == '>') { ++c.i; break; }
      if (c.StartsWith("/>")) { c.i += 2; selfClosed = true; break; }
      size_t ab = c.i;
      while (!c.Eof() && IsNameChar(text[c.i])) ++c.i;
      if (c.i == ab) return fail("malformed attribute");
      std::string an = text.substr(ab, c.i - ab);
      c.SkipSpace();
      if (c.Eof() || text[c.i] != '=') return fail("attribute without value");
      ++c.i;
      c.SkipSpace();
      if (c.Eof() || (text[c.i] != '"' && text[c.i] != '\'')) return fail("unquoted attribute value");
      const char q = text[c.i++];
      size_t vb = c.i;
      size_t ve = text.find(q, vb);
      if (ve == std::string::npos) return fail("unterminated attribute value");
      el->attrs_.emplace_back(std::move(an), DecodeEntities(text.substr(vb, ve - vb)));
      c.i = ve + 1;
    }
    XmlElement *raw = el.get();
    auto &siblings = open.empty() ? roots_ : open.back()->children_;
    if (!siblings.empty()) siblings.back()->next_ = raw;
    siblings.push_back(std::move(el));
    if (!selfClosed) open.push_back(raw);
  }
  if (!open.empty()) return fail("unclosed element <" + open.back()->name_ + ">");
  return true;
}

}  // namespace qaray_hip
