// Settings.h -- what the cipher path needs of the reference's global settings
// (Modulate/Settings.h:5-23, Settings.cpp:4-9): which platform's magic/key pair applies, and the
// output-policy switches SaveArk / ExtractFiles consult.  Header-only (C++17 inline statics).
//
// The names the reference's code uses (CSettings::mbPS4, msPlatform, kuEncrypted*...) are kept so
// call sites read the same; the values live in one platform table.
#pragma once

#include <iostream>

class CSettings
{
public:
    struct SPlatform
    {
        const char* mpName;   // "ps3" / "ps4": main_<name>.hdr, main_<name>_<i>.ark
        unsigned int muMagic; // plaintext LE u32 at offset 0 of an encrypted header
        unsigned int muKey;   // initial LCG key for that platform's headers
    };
    // Settings.h:16-20 -- PS3 uses its magic as the key; PS4 has a separate key
    static constexpr SPlatform kaPlatforms[ 2 ] = {
        { "ps3", 0xc64eed30u, 0xc64eed30u },
        { "ps4", 0x6f303f55u, 0x90cfc0abu },
    };
    static constexpr unsigned int kuEncryptedVersionPS3 = kaPlatforms[ 0 ].muMagic;
    static constexpr unsigned int kuEncryptedVersionPS4 = kaPlatforms[ 1 ].muMagic;
    static constexpr unsigned int kuEncryptedPS3Key = kaPlatforms[ 0 ].muKey;
    static constexpr unsigned int kuEncryptedPS4Key = kaPlatforms[ 1 ].muKey;

    // Save side: chosen by the -ps3 switch (CArk.cpp:914, 1136).  Load side: by the file's magic.
    inline static bool mbPS4 = true;
    inline static const char* msPlatform = kaPlatforms[ 1 ].mpName;
    static void SelectPlatform( bool lbPS4 )
    {
        mbPS4 = lbPS4;
        msPlatform = kaPlatforms[ lbPS4 ? 1 : 0 ].mpName;
    }
    static const SPlatform& Current() { return kaPlatforms[ mbPS4 ? 1 : 0 ]; }
    // nullptr if luMagic is neither platform's (eError_UnknownVersionNumber at the call sites)
    static const SPlatform* FromMagic( unsigned int luMagic )
    {
        for( const SPlatform& p : kaPlatforms )
            if( p.muMagic == luMagic ) return &p;
        return nullptr;
    }

    // defaults as Settings.cpp:6-9: quiet, overwrite outputs, ignore files the reference header
    // does not know, apply the song filter
    inline static bool mbVerbose = false;
    inline static bool mbOverwriteOutputFiles = true;
    inline static bool mbIgnoreNewFiles = true;
    inline static bool mbPackAllFiles = false;

    // Addition.  Off (the default) reproduces every deterministic behaviour of the reference, its
    // quirks included; on (-fixquirks) switches all of these to their corrected forms at once:
    //   * SaveArk keeps the slice pointer where it was after a part it did not write, so later
    //     parts get the wrong bytes (CArk.cpp:866, 883-891)      -> fixed: slices follow part sizes
    //   * ExtractFiles walks every entry whatever its two index arguments say (CArk.cpp:435)
    //                                                             -> fixed: the range is honoured
    //   * SaveArk refuses to run unless lpHeaderFilename can be opened in the working directory
    //     (CArk.cpp:904-909)                                      -> fixed: no such requirement
    //   * CDtaFile::Save writes top-level trees back to back, which its own Load cannot read when
    //     there are several (CDtaFile.cpp:371-374)                -> fixed: separators written
    inline static bool mbFixReferenceQuirks = false;
};

#define VERBOSE_OUT( out ) do { if( CSettings::mbVerbose ) std::cout << out; } while( 0 )
