// Commands.h -- the three reference commands that sit on the cipher path, for Linux:
//   Decode  Modulate/Modulate.cpp:452-502      Unpack  Modulate/Modulate.cpp:291-317
//   Pack    Modulate/Modulate.cpp:380-450  (with -packall semantics: no song-list filter, which
//           needs the out-of-scope CDtaFile)
// The reference finds main_<platform>.hdr in the working directory; these take the directory.
#pragma once
#include <string>
#include "Error.h"

eError Decode( const std::string& lDirectory );
eError Unpack( const std::string& lHeaderDirectory, const std::string& lOutputDirectory, bool lbCryptParts, int liNumDevices );
eError Pack( const std::string& lHeaderDirectory, const std::string& lInputDirectory, const std::string& lOutputDirectory, bool lbCryptParts, int liNumDevices );
