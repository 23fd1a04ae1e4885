// Settings.h -- the slice of Modulate/Settings.h:5-23 the cipher path needs: platform selector,
// header magics and keys (Settings.h:16-20), and the output flags SaveArk/ExtractFiles consult.
#pragma once

#include <iostream>

class CSettings
{
public:
    static bool mbPS4;                  // save-side platform (CArk.cpp:914, 1136); load picks by magic
    static const char* msPlatform;      // "ps4" / "ps3": main_<platform>.hdr
    static bool mbVerbose;
    static bool mbOverwriteOutputFiles; // CArk.cpp:853-869, 443-457
    static bool mbIgnoreNewFiles;
    static bool mbPackAllFiles;

    static constexpr unsigned int kuEncryptedVersionPS3 = 0xc64eed30u; // header magic, PS3
    static constexpr unsigned int kuEncryptedVersionPS4 = 0x6f303f55u; // header magic, PS4
    static constexpr unsigned int kuEncryptedPS3Key = 0xc64eed30u;     // cipher key, PS3
    static constexpr unsigned int kuEncryptedPS4Key = 0x90cfc0abu;     // cipher key, PS4

    static void SelectPlatform( bool lbPS4 )
    {
        mbPS4 = lbPS4;
        msPlatform = lbPS4 ? "ps4" : "ps3";
    }
};

#define VERBOSE_OUT( out ) do { if( CSettings::mbVerbose ) std::cout << out; } while( 0 )
