// stub (see ../../README.md): the viewer is out of scope; drawer.h only names this type
#pragma once
namespace pangolin { struct OpenGlMatrix { double m[16]; void SetIdentity(); }; }
