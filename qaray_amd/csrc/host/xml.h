// xml.h — small XML DOM reader for qaray scene files.
//
// The reference reads its scenes through TinyXML 2.6.2 (external/tinyxml) and only uses a sliver
// of it: element names, string/double/int attributes, first-child / next-sibling iteration
// (src/parser/xmlload.cpp:71-149).  This is an independent reader covering that sliver:
// elements, attributes, comments, declarations and CDATA/text skipping, standard entities.
#pragma once
#include <memory>
#include <string>
#include <utility>
#include <vector>

namespace qaray_hip {

class XmlElement {
 public:
  const std::string &Value() const { return name_; }
  // nullptr when the attribute is absent (TiXmlElement::Attribute)
  const char *Attribute(const char *name) const;
  // Leave *out untouched when absent or unparsable (TiXmlElement::QueryDoubleAttribute, which
  // converts with sscanf("%lf")); returns true when a value was stored.
  bool QueryDoubleAttribute(const char *name, double *out) const;
  bool QueryIntAttribute(const char *name, int *out) const;
  const XmlElement *FirstChildElement() const { return children_.empty() ? nullptr : children_.front().get(); }
  const XmlElement *FirstChildElement(const char *name) const;
  const XmlElement *NextSiblingElement() const { return next_; }
  const std::vector<std::unique_ptr<XmlElement>> &Children() const { return children_; }

 private:
  friend class XmlDocument;
  std::string name_;
  std::vector<std::pair<std::string, std::string>> attrs_;
  std::vector<std::unique_ptr<XmlElement>> children_;
  const XmlElement *next_ = nullptr;
};

class XmlDocument {
 public:
  // Returns false (and sets Error()) on I/O or syntax errors.
  bool LoadFile(const char *filename);
  bool Parse(const std::string &text);
  const XmlElement *FirstChildElement(const char *name) const;
  const std::string &Error() const { return error_; }

 private:
  std::vector<std::unique_ptr<XmlElement>> roots_;
  std::string error_;
};

}  // namespace qaray_hip
