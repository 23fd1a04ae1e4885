// Settings.cpp -- defaults as Modulate/Settings.cpp:4-9 (PS4, overwrite on, quiet).
#include "Settings.h"

bool CSettings::mbPS4 = true;
const char* CSettings::msPlatform = "ps4";
bool CSettings::mbVerbose = false;
bool CSettings::mbOverwriteOutputFiles = true;
bool CSettings::mbIgnoreNewFiles = true;
bool CSettings::mbPackAllFiles = false;
