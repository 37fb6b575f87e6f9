// ungar_amd :: optional Boost.Hana hook.
#pragma once

// User code written against the reference may use Boost.Hana through the alias the reference exports
// (`hana::unpack(Utils::Decompose<...>(v), ...)`, utils.test.cpp:118-137).  The facade itself does not need Hana; when the
// header is on the include path the alias is provided.
#if defined(__has_include)
#if __has_include(<boost/hana.hpp>)
#include <boost/hana.hpp>
#include <boost/hana/ext/std/array.hpp>
#include <boost/hana/ext/std/tuple.hpp>
#define UNGAR_AMD_HAS_HANA 1
#endif
#endif


#if defined(UNGAR_AMD_HAS_HANA)
namespace Ungar {
namespace hana = boost::hana;
}
#endif
