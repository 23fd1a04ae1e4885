// modulate_main.cpp -- a small Linux command line over the cipher path only:
//   modulate [-ps3] [-verbose] [-force] [-packall] [-gpus N] [-cryptparts] [-fixquirks]
//            -decode <dir> | -unpack <hdr_dir> <out_dir> | -pack <hdr_dir> <in_dir> <out_dir>
// Flag names and their order-sensitivity follow Modulate/Modulate.cpp:895-972 (flags act when
// reached).  The song / DTA commands of the reference are out of scope.
// -gpus, -cryptparts and -fixquirks are additions: -cryptparts switches on the part-level cipher
// (BASELINE.json north_star; without it parts are stored raw exactly as the reference does),
// -fixquirks the corrected forms of the reference's deterministic quirks (Settings.h).
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <exception>
#include <iostream>
#include <string>
#include <strings.h>

#include "Commands.h"
#include "Settings.h"

int main( int argc, char* argv[] )
{
    std::deque< std::string > laParams( argv + 1, argv + argc );
    bool lbCryptParts = false;
    int liNumDevices = 0;
    if( laParams.empty() )
    {
        std::cout << "usage: modulate [-ps3] [-verbose] [-force] [-packall] [-gpus N] [-cryptparts] [-fixquirks] -decode <dir> | -unpack <hdr_dir> <out> | -pack <hdr_dir> <in> <out>\n";
        return 0;
    }
    auto lPop = [ & ]( std::string& lOut ) {
        if( laParams.empty() ) return false;
        lOut = laParams.front();
        laParams.pop_front();
        return true;
    };
    try
    {
        while( !laParams.empty() )
        {
            std::string lCmd;
            lPop( lCmd );
            eError leError = eError_NoError;
            std::string a, b, c;
            if( !strcasecmp( lCmd.c_str(), "-ps3" ) ) { CSettings::SelectPlatform( false ); std::cout << "Switching to PS3 mode\n"; }
            else if( !strcasecmp( lCmd.c_str(), "-verbose" ) ) { CSettings::mbVerbose = true; std::cout << "Verbose mode enabled\n"; }
            else if( !strcasecmp( lCmd.c_str(), "-force" ) ) CSettings::mbOverwriteOutputFiles = true;
            else if( !strcasecmp( lCmd.c_str(), "-packall" ) ) { CSettings::mbPackAllFiles = true; CSettings::mbIgnoreNewFiles = false; }
            else if( !strcasecmp( lCmd.c_str(), "-cryptparts" ) ) lbCryptParts = true;
            else if( !strcasecmp( lCmd.c_str(), "-fixquirks" ) ) CSettings::mbFixReferenceQuirks = true;
            else if( !strcasecmp( lCmd.c_str(), "-gpus" ) ) leError = lPop( a ) ? ( liNumDevices = std::atoi( a.c_str() ), eError_NoError ) : eError_InvalidParameter;
            else if( !strcasecmp( lCmd.c_str(), "-decode" ) ) leError = lPop( a ) ? Decode( a ) : eError_InvalidParameter;
            else if( !strcasecmp( lCmd.c_str(), "-unpack" ) ) leError = ( lPop( a ) && lPop( b ) ) ? Unpack( a, b, lbCryptParts, liNumDevices ) : eError_InvalidParameter;
            else if( !strcasecmp( lCmd.c_str(), "-pack" ) ) leError = ( lPop( a ) && lPop( b ) && lPop( c ) ) ? Pack( a, b, c, lbCryptParts, liNumDevices ) : eError_InvalidParameter;
            else
            {
                std::cout << "Unkown parameter: " << lCmd << "\nAborting\n\n";
                return -1;
            }
            if( leError != eError_NoError )
            {
                ShowError( leError );
                return -1;
            }
            std::cout << "\n";
        }
    }
    catch( const std::exception& e )
    {
        std::cout << "ERROR: " << e.what() << "\n";
        return -2;
    }
    std::cout << "Complete!\n";
    return 0;
}
