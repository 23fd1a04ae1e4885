// Commands.cpp -- see Commands.h.
#include "Commands.h"

#include <cstdio>
#include <filesystem>
#include <iostream>
#include <vector>

#include "CArk.h"
#include "CEncryptionCycler.h"
#include "Settings.h"

namespace
{
std::string Slashed( std::string s )
{
    if( !s.empty() && s.back() != '/' && s.back() != '\\' ) s += '/';
    return s;
}
std::string HeaderName() { return std::string( "main_" ) + CSettings::msPlatform + ".hdr"; }

std::string Absolute( const std::string& lPath )
{
    std::error_code ec;
    std::filesystem::path p = std::filesystem::absolute( lPath, ec );
    return ec ? lPath : p.string();
}
} // namespace

eError Decode( const std::string& lDirectory ) // Modulate.cpp:452-502
{
    const std::string lHeaderFilename = Slashed( lDirectory ) + HeaderName();
    VERBOSE_OUT( "Loading header file " << lHeaderFilename );
    FILE* f = std::fopen( lHeaderFilename.c_str(), "rb" );
    if( !f ) return eError_FailedToOpenFile;
    std::fseek( f, 0, SEEK_END );
    long n = std::ftell( f );
    std::fseek( f, 0, SEEK_SET );
    std::vector< unsigned char > lData( n > 0 ? (size_t)n : 0 );
    size_t got = lData.empty() ? 0 : std::fread( lData.data(), 1, lData.size(), f );
    std::fclose( f );
    VERBOSE_OUT( "\nLoaded header (" << lData.size() << ") bytes\n" );
    if( got != lData.size() || lData.size() < 4 ) return eError_UnknownVersionNumber;

    const unsigned int luVersion = (unsigned int)lData[ 0 ] | ( (unsigned int)lData[ 1 ] << 8 ) | ( (unsigned int)lData[ 2 ] << 16 ) | ( (unsigned int)lData[ 3 ] << 24 );
    if( luVersion != CSettings::kuEncryptedVersionPS3 && luVersion != CSettings::kuEncryptedVersionPS4 ) return eError_UnknownVersionNumber;
    const unsigned int kuInitialKey = luVersion == CSettings::kuEncryptedVersionPS3 ? CSettings::kuEncryptedPS3Key : CSettings::kuEncryptedPS4Key;

    CEncryptionCycler lDecrypt;
    lDecrypt.Cycle( lData.data() + 4, (unsigned int)( lData.size() - 4 ), (int)kuInitialKey ); // Modulate.cpp:485-486

    f = std::fopen( ( lHeaderFilename + ".dec" ).c_str(), "wb" );
    if( !f ) return eError_FailedToCreateFile;
    size_t w = std::fwrite( lData.data(), 1, lData.size(), f );
    std::fclose( f );
    return w == lData.size() ? eError_NoError : eError_FailedToWriteData;
}

eError Unpack( const std::string& lHeaderDirectory, const std::string& lOutputDirectory, bool lbCryptParts, int liNumDevices ) // Modulate.cpp:291-317
{
    std::cout << "Unpacking " << HeaderName() << " to " << lOutputDirectory << "\n";
    CArk lArkHeader;
    lArkHeader.EnablePartCipher( lbCryptParts, liNumDevices ); // off = reference behaviour
    eError leError = lArkHeader.Load( ( Slashed( lHeaderDirectory ) + HeaderName() ).c_str() );
    SHOW_ERROR_AND_RETURN;
    return lArkHeader.ExtractFiles( 0, lArkHeader.GetNumFiles(), Slashed( lOutputDirectory ).c_str() );
}

eError Pack( const std::string& lHeaderDirectory, const std::string& lInputDirectory, const std::string& lOutputDirectory, bool lbCryptParts, int liNumDevices ) // Modulate.cpp:380-450
{
    std::cout << "Packing " << HeaderName() << " from " << lInputDirectory << " to " << lOutputDirectory << "\n";
    // The reference works in the directory that holds main_<platform>.hdr: it opens the header, and SaveArk re-opens it,
    // by bare file name (Modulate.cpp:383-395, CArk.cpp:904-909).  These commands take that directory as an argument;
    // it is handed to the two CArk objects as their working directory -- the process's own is never changed (this code
    // is reachable through host_capi from multi-threaded hosts).
    const std::string lInput = Slashed( Absolute( lInputDirectory ) ), lOutput = Slashed( Absolute( lOutputDirectory ) );
    const std::string lWork = Slashed( lHeaderDirectory );
    CArk lReferenceArkHeader;
    eError leError = lReferenceArkHeader.Load( ( lWork + HeaderName() ).c_str() );
    SHOW_ERROR_AND_RETURN;
    CArk lArkHeader;
    lArkHeader.SetWorkingDirectory( lWork );
    leError = lArkHeader.ConstructFromDirectory( lInput.c_str(), lReferenceArkHeader, {} );
    SHOW_ERROR_AND_RETURN;
    leError = lArkHeader.BuildArk( lInput.c_str(), {} );
    SHOW_ERROR_AND_RETURN; // (the reference ignores the status of these two calls, Modulate.cpp:445-446)
    lArkHeader.EnablePartCipher( lbCryptParts, liNumDevices );
    return lArkHeader.SaveArk( lOutput.c_str(), HeaderName().c_str() );
}
