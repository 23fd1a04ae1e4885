// CArk.cpp -- see CArk.h.  Written for Linux/POSIX from the behaviour of Modulate/CArk.cpp (cited
// per function); not a translation: the reference is Win32-only and pointer-walks
// fixed-size buffers, this keeps bounds-checked vectors and lifts its size limits (F5).
#include "CArk.h"

#include <algorithm>
#include <atomic>
#include <cctype>
#include <cstdio>
#include <cstring>
#include <filesystem>
#include <thread>
#include <unordered_map>

#include "../../../include/modgpu.h"
#include "CEncryptionCycler.h"
#include "Settings.h"

namespace fs = std::filesystem;

namespace
{
constexpr unsigned int kuUnencryptedVersion = 9; // CArk.cpp:309, 903
constexpr int kiMaxArks = 4096;                  // reference: 100 (CArk.cpp:345); lifted
constexpr int kiMaxFiles = 1 << 24;              // reference: 25 000 (CArk.cpp:395); lifted (F5)
constexpr int kiMaxStringLength = 255;           // CArk.cpp:545, 610

// ---- little-endian byte sink / source (the reference casts pointers into a raw buffer)
struct ByteSink
{
    std::vector< unsigned char >& m;
    void U32( uint32_t v ) { for( int i = 0; i < 4; ++i ) m.push_back( (unsigned char)( v >> ( 8 * i ) ) ); }
    void I64( int64_t v ) { for( int i = 0; i < 8; ++i ) m.push_back( (unsigned char)( (uint64_t)v >> ( 8 * i ) ) ); }
    void Str( const std::string& s ) { U32( (uint32_t)s.size() ); m.insert( m.end(), s.begin(), s.end() ); }
    void Zero( size_t n ) { m.insert( m.end(), n, 0 ); }
};

struct ByteSource
{
    const unsigned char* p;
    size_t n, at = 0;
    bool ok = true;
    bool Need( size_t k ) { if( !ok || n - at < k ) ok = false; return ok; }
    uint32_t U32() { if( !Need( 4 ) ) return 0; uint32_t v = 0; for( int i = 0; i < 4; ++i ) v |= (uint32_t)p[ at + i ] << ( 8 * i ); at += 4; return v; }
    int64_t I64() { if( !Need( 8 ) ) return 0; uint64_t v = 0; for( int i = 0; i < 8; ++i ) v |= (uint64_t)p[ at + i ] << ( 8 * i ); at += 8; return (int64_t)v; }
    void Skip( size_t k ) { if( Need( k ) ) at += k; }
    std::string Str() // length-prefixed; the value is clipped to 255 chars but the cursor moves the full length (CArk.cpp:604-621)
    {
        int32_t len = (int32_t)U32();
        if( len < 0 || !Need( (size_t)len ) ) { ok = false; return std::string(); }
        std::string s( (const char*)p + at, (size_t)std::min( len, kiMaxStringLength ) );
        at += (size_t)len;
        size_t z = s.find( '\0' ); // the reference builds the value through a C string
        if( z != std::string::npos ) s.resize( z );
        return s;
    }
};

// File-name bucket hash of the header's lookup table (CArk.cpp:832-843): signed-char, int arithmetic.
int NameBucket( const std::string& lName, int liNumFiles )
{
    int h = 0;
    const char* p = lName.c_str();
    do
    {
        h = h * 0x7F + *p;
        h -= ( h / liNumFiles ) * liNumFiles;
    } while( *( ++p ) );
    return h;
}

std::vector< std::string > LowerComponents( const std::string& lName )
{
    std::vector< std::string > out( 1 );
    for( char c : lName )
    {
        if( c == '/' ) out.emplace_back();
        else out.back().push_back( (char)std::tolower( (unsigned char)c ) );
    }
    return out;
}

bool ReadWholeFile( const std::string& lPath, std::vector< unsigned char >& lOut )
{
    FILE* f = std::fopen( lPath.c_str(), "rb" );
    if( !f ) return false;
    std::fseek( f, 0, SEEK_END );
    long n = std::ftell( f );
    std::fseek( f, 0, SEEK_SET );
    lOut.resize( n > 0 ? (size_t)n : 0 );
    size_t got = lOut.empty() ? 0 : std::fread( lOut.data(), 1, lOut.size(), f );
    std::fclose( f );
    return got == lOut.size();
}

// Existing non-empty output is kept when overwriting is off (CArk.cpp:443-457, 853-869).
bool KeepExisting( const std::string& lPath )
{
    if( CSettings::mbOverwriteOutputFiles ) return false;
    std::error_code ec;
    auto sz = fs::file_size( lPath, ec );
    return !ec && sz != 0;
}

eError WriteWholeFile( const std::string& lPath, const void* lpData, size_t n )
{
    FILE* f = std::fopen( lPath.c_str(), "wb" );
    if( !f ) return eError_FailedToCreateFile;
    size_t w = n ? std::fwrite( lpData, 1, n, f ) : 0;
    std::fclose( f );
    return w == n ? eError_NoError : eError_FailedToWriteData;
}

std::string WithSlash( const char* lpDir )
{
    std::string s = lpDir ? lpDir : "";
    if( !s.empty() && s.back() != '/' ) s += '/';
    return s;
}
} // namespace

// The reference walks its file table on one thread; the per-entry file I/O here is independent, so it is
// spread over a few host threads (entries are claimed through an atomic cursor).  lWork returns an eError;
// the first failure stops the walk.
template < typename F > eError ParallelOverEntries( int liBegin, int liEnd, F lWork )
{
    const int liThreads = (int)std::min< int64_t >( std::max( 1u, std::min( 16u, std::thread::hardware_concurrency() ) ), std::max( 1, ( liEnd - liBegin ) / 64 ) );
    std::atomic< int > lNext( liBegin );
    std::atomic< int > lFirstError( (int)eError_NoError );
    auto lBody = [ & ] {
        for( ;; )
        {
            const int ii = lNext.fetch_add( 1 );
            if( ii >= liEnd || lFirstError.load() != (int)eError_NoError ) return;
            eError e = lWork( ii );
            int liExpected = (int)eError_NoError;
            if( e != eError_NoError ) lFirstError.compare_exchange_strong( liExpected, (int)e );
        }
    };
    std::vector< std::thread > lThreads;
    for( int t = 1; t < liThreads; ++t ) lThreads.emplace_back( lBody );
    lBody();
    for( std::thread& t : lThreads ) t.join();
    return (eError)lFirstError.load();
}

CArkDataBuffer::~CArkDataBuffer() { Release(); }

void CArkDataBuffer::Release()
{
    if( mpData ) modgpu_host_free( mpData );
    mpData = nullptr;
    muSize = 0;
}

bool CArkDataBuffer::Allocate( uint64_t luSize, const std::vector< uint64_t >& laPartSizes, int liNumDevices )
{
    Release();
    void* lp = nullptr;
    uint64_t luSum = 0;
    for( uint64_t luPart : laPartSizes ) luSum += luPart;
    // With the part plan known, the buffer is still ONE contiguous allocation (the reference's layout, CArk.cpp:738, 780),
    // but each part's pages sit on the NUMA node of the GPU the part cipher will send it to (part i -> GPU i mod N), and
    // the pages are first-touched in parallel before they are locked -- half the time of one hipHostMalloc of the whole.
    const int liStatus = ( !laPartSizes.empty() && luSum == luSize )
                             ? modgpu_host_alloc_parts( &lp, laPartSizes.data(), (int)laPartSizes.size(), liNumDevices )
                             : modgpu_host_alloc( &lp, luSize );
    if( liStatus != MODGPU_OK ) return false;
    mpData = static_cast< char* >( lp );
    muSize = luSize;
    return true;
}

std::vector< uint64_t > CArk::PartSizes() const
{
    std::vector< uint64_t > laSizes;
    for( const sArkDefinition& a : maArks ) laSizes.push_back( a.muSize );
    return laSizes;
}

CArk::CArk() = default;
CArk::~CArk() = default;

int CArk::GetNumFiles() const { return (int)maFiles.size(); }
bool CArk::IsArkDataPinned() const { return maArkData.size() != 0 && modgpu_host_is_pinned( maArkData.data(), maArkData.size() ) != 0; }

const CArk::sFileDefinition* CArk::GetFile( const std::string& lName ) const
{
    for( const sFileDefinition& f : maFiles )
        if( f.mName == lName ) return &f;
    return nullptr;
}

bool CArk::FileExists( const char* lpFilename ) const // CArk.cpp:1210-1225
{
    return lpFilename && GetFile( lpFilename ) != nullptr;
}

// ------------------------------------------------------------------------------------ Load
eError CArk::Load( const char* lpHeaderFilename ) // CArk.cpp:301-422
{
    if( mbLoaded ) return eError_AlreadyLoaded;
    VERBOSE_OUT( "Loading header file " << lpHeaderFilename );
    std::vector< unsigned char > lImage;
    if( !lpHeaderFilename || !ReadWholeFile( lpHeaderFilename, lImage ) ) return eError_FailedToOpenFile;
    VERBOSE_OUT( "\nLoaded header (" << lImage.size() << ") bytes\n" );
    mHeaderDirectory = fs::path( lpHeaderFilename ).parent_path().string();
    if( !mHeaderDirectory.empty() ) mHeaderDirectory += '/';
    return ParseHeader( std::move( lImage ) );
}

eError CArk::ParseHeader( std::vector< unsigned char > lImage )
{
    if( lImage.size() < 4 ) return eError_UnknownVersionNumber;
    ByteSource lMagic{ lImage.data(), lImage.size() };
    const unsigned int luVersion = lMagic.U32(); // plaintext LE u32 at offset 0 (CArk.cpp:328)
    if( luVersion != CSettings::kuEncryptedVersionPS3 && luVersion != CSettings::kuEncryptedVersionPS4 )
        return eError_UnknownVersionNumber;
    // on load the key follows the file's magic, not the -ps3 switch (CArk.cpp:336)
    const unsigned int kuInitialKey = luVersion == CSettings::kuEncryptedVersionPS3 ? CSettings::kuEncryptedPS3Key : CSettings::kuEncryptedPS4Key;

    CEncryptionCycler lDecrypt;
    lDecrypt.Cycle( lImage.data() + 4, (unsigned int)( lImage.size() - 4 ), (int)kuInitialKey ); // CArk.cpp:338-339

    ByteSource in{ lImage.data(), lImage.size(), 4 };
    in.U32();      // muVersion (9)
    in.U32();      // miNumChecksums
    in.Skip( 16 ); // mChecksumData
    const int liNumArks = (int)in.U32();
    if( !in.ok ) { eError leError = eError_InvalidData; SHOW_ERROR_AND_RETURN; }
    if( liNumArks < 0 || liNumArks > kiMaxArks ) { eError leError = eError_ValueOutOfBounds; SHOW_ERROR_AND_RETURN; }

    std::vector< sArkDefinition > lArks( (size_t)liNumArks );
    const int liNumSizes = (int)in.U32(); // sIntList arkSizes (CArk.cpp:354-364)
    for( int ii = 0; ii < liNumSizes && in.ok; ++ii )
    {
        unsigned int luSize = in.U32();
        if( ii < liNumArks ) lArks[ ii ].muSize = luSize;
    }
    if( !in.ok || liNumSizes < liNumArks ) { eError leError = eError_ValueOutOfBounds; SHOW_ERROR_AND_RETURN; }
    const int liNumPaths = (int)in.U32(); // sStringList arkPaths (CArk.cpp:366-376)
    for( int ii = 0; ii < liNumPaths && in.ok; ++ii )
    {
        std::string lPath = in.Str();
        if( ii < liNumArks ) lArks[ ii ].mPath = lPath;
    }
    if( !in.ok || liNumPaths < liNumArks ) { eError leError = eError_ValueOutOfBounds; SHOW_ERROR_AND_RETURN; }
    const int liNumChecksums = (int)in.U32(); // CArk.cpp:378-390
    if( liNumChecksums < 0 ) { eError leError = eError_InvalidData; SHOW_ERROR_AND_RETURN; }
    in.Skip( (size_t)liNumChecksums * 4 ); // checksums
    in.Skip( (size_t)liNumChecksums * 4 ); // "hashes"
    if( in.U32() != 0 || !in.ok ) { eError leError = eError_InvalidData; SHOW_ERROR_AND_RETURN; }

    const int liNumFiles = (int)in.U32(); // CArk.cpp:392-399
    if( !in.ok || liNumFiles < 0 || liNumFiles > kiMaxFiles ) { eError leError = eError_ValueOutOfBounds; SHOW_ERROR_AND_RETURN; }
    std::vector< sFileDefinition > lFiles( (size_t)liNumFiles );
    for( sFileDefinition& f : lFiles ) // sFileDefinition::InitialiseFromData (CArk.cpp:594-650)
    {
        f.mi64Offset = in.I64();
        f.mName = in.Str();
        f.miFlags1 = (int)in.U32();
        f.miSize = (int)in.U32();
        f.miHash = (int)in.U32();
        if( !in.ok || f.mName.empty() || f.mi64Offset < 0 || f.miSize < 0 ) { eError leError = eError_InvalidData; SHOW_ERROR_AND_RETURN; }
    }
    const int liNumFlags2 = (int)in.U32(); // trailing sIntList, read back as miFlags2 (CArk.cpp:410-416)
    for( int ii = 0; ii < liNumFiles; ++ii )
    {
        if( ii >= liNumFlags2 ) { eError leError = eError_ValueOutOfBounds; SHOW_ERROR_AND_RETURN; }
        lFiles[ ii ].miFlags2 = (int)in.U32();
    }
    if( !in.ok ) { eError leError = eError_InvalidData; SHOW_ERROR_AND_RETURN; }

    maArks = std::move( lArks );
    maFiles = std::move( lFiles );
    miLoadedKey = (int)kuInitialKey;
    mbLoaded = true;
    return eError_NoError;
}

// ------------------------------------------------------------------------------------ parts in
// Runs lWork( part index, device ) for every part, part i on GPU i mod N, one host thread per GPU
// (no inter-GPU traffic: parts are independent streams).  Returns the first failure.
// On a host WITHOUT a GPU the part cipher is still defined -- it is Cycle over a part, and the reference's Cycle cannot
// fail (CEncryptionCycler.cpp:4-14) -- so the parts go, one after another, through the library's own host loop
// (lWork( i, -1 ): device -1 = no device; modgpu_cycle_auto_host, which threads large buffers itself).  This is "the build's
// CPU path" BASELINE config 5 byte-diffs the GPU output against.  MODGPU_REQUIRE_GPU=1 forbids it, as everywhere.
template < typename F > static eError ForEachPartOnDevices( size_t liNumParts, int liNumDevices, F lWork )
{
    int liAvailable = modgpu_device_count();
    if( liAvailable <= 0 )
    {
        if( modgpu_gpu_required() )
        {
            std::printf( "ERROR: GPU part cipher failed: no HIP device visible\n" );
            return eError_InvalidData;
        }
        for( size_t i = 0; i < liNumParts; ++i )
        {
            eError leError = lWork( i, -1 );
            if( leError != eError_NoError ) return leError;
        }
        return eError_NoError;
    }
    if( liNumDevices <= 0 || liNumDevices > liAvailable ) liNumDevices = liAvailable;
    liNumDevices = (int)std::min< size_t >( (size_t)liNumDevices, std::max< size_t >( liNumParts, 1 ) );
    std::vector< eError > lResults( (size_t)liNumDevices, eError_NoError );
    std::vector< std::thread > lThreads;
    for( int d = 0; d < liNumDevices; ++d )
        lThreads.emplace_back( [ &, d ] {
            for( size_t i = (size_t)d; i < liNumParts && lResults[ d ] == eError_NoError; i += (size_t)liNumDevices )
                lResults[ d ] = lWork( i, d );
        } );
    for( std::thread& t : lThreads ) t.join();
    for( eError e : lResults )
        if( e != eError_NoError ) return e;
    return eError_NoError;
}

eError CArk::LoadArkData() // CArk.cpp:723-758
{
    uint64_t luTotalArkSize = 0;
    std::vector< uint64_t > lOffsets;
    std::vector< std::string > lPaths;
    for( const sArkDefinition& a : maArks )
    {
        lOffsets.push_back( luTotalArkSize );
        luTotalArkSize += a.muSize;
        // as the reference: relative to the working directory; else beside the header that was loaded
        std::error_code ec;
        lPaths.push_back( fs::exists( a.mPath, ec ) ? a.mPath : mHeaderDirectory + a.mPath );
        if( !fs::is_regular_file( lPaths.back(), ec ) ) { eError leError = eError_FailedToOpenFile; SHOW_ERROR_AND_RETURN; }
        if( fs::file_size( lPaths.back(), ec ) < a.muSize ) { eError leError = eError_InvalidData; SHOW_ERROR_AND_RETURN; } // the reference does not check (CArk.cpp:751)
    }
    if( !maArkData.Allocate( luTotalArkSize, PartSizes(), miPartDevices ) ) { eError leError = eError_NoData; SHOW_ERROR_AND_RETURN; } // CArk.cpp:738
    if( !mbPartCipher ) // the reference's behaviour: parts are stored raw
    {
        for( size_t ii = 0; ii < maArks.size(); ++ii )
        {
            FILE* f = std::fopen( lPaths[ ii ].c_str(), "rb" );
            if( !f ) { eError leError = eError_FailedToOpenFile; SHOW_ERROR_AND_RETURN; }
            size_t got = maArks[ ii ].muSize ? std::fread( maArkData.data() + lOffsets[ ii ], 1, maArks[ ii ].muSize, f ) : 0;
            std::fclose( f );
            if( got != maArks[ ii ].muSize ) { eError leError = eError_InvalidData; SHOW_ERROR_AND_RETURN; }
        }
        return eError_NoError;
    }
    // addition: parts are ciphertext on disk.  Each part file is streamed disk -> GPU -> its slice of
    // the buffer (pread / H2D / kernel / D2H overlapped, modgpu_cycle_file_to_host), part i on GPU i mod N.
    const int liKey = miLoadedKey ? miLoadedKey : (int)CSettings::Current().muKey;
    eError leError = ForEachPartOnDevices( maArks.size(), miPartDevices, [ & ]( size_t ii, int liDevice ) {
        if( liDevice < 0 ) // no GPU on this host: read the part as the reference does (CArk.cpp:751), then Cycle's host engine
        {
            uint8_t* lpSlice = reinterpret_cast< uint8_t* >( maArkData.data() ) + lOffsets[ ii ];
            FILE* f = std::fopen( lPaths[ ii ].c_str(), "rb" );
            if( !f ) return eError_FailedToOpenFile;
            size_t got = maArks[ ii ].muSize ? std::fread( lpSlice, 1, maArks[ ii ].muSize, f ) : 0;
            std::fclose( f );
            if( got != maArks[ ii ].muSize ) return eError_InvalidData;
            return modgpu_cycle_auto_host( lpSlice, maArks[ ii ].muSize, liKey, 0, -1 ) == MODGPU_OK ? eError_NoError : eError_InvalidData;
        }
        int liStatus = modgpu_cycle_file_to_host( lPaths[ ii ].c_str(), 0, reinterpret_cast< uint8_t* >( maArkData.data() ) + lOffsets[ ii ],
                                                  maArks[ ii ].muSize, liKey, 0, liDevice );
        if( liStatus == MODGPU_OK ) return eError_NoError;
        std::printf( "ERROR: part %s: %s\n", lPaths[ ii ].c_str(), modgpu_last_error() );
        return liStatus == MODGPU_ERR_IO ? eError_FailedToOpenFile : eError_InvalidData;
    } );
    SHOW_ERROR_AND_RETURN;
    return eError_NoError;
}

eError CArk::CycleArkData( int liKey, int liNumDevices ) // addition: north_star part cipher (in place, in memory)
{
    std::vector< uint8_t* > lParts;
    std::vector< uint64_t > lSizes;
    uint64_t luOffset = 0;
    for( const sArkDefinition& a : maArks )
    {
        if( luOffset + a.muSize > maArkData.size() ) return eError_NoData;
        lParts.push_back( reinterpret_cast< uint8_t* >( maArkData.data() ) + luOffset );
        lSizes.push_back( a.muSize );
        luOffset += a.muSize;
    }
    if( lParts.empty() ) return eError_NoError;
    if( modgpu_device_count() <= 0 && !modgpu_gpu_required() ) // no GPU on this host: every part through Cycle's host engine
    {
        for( size_t i = 0; i < lParts.size(); ++i )
            if( modgpu_cycle_auto_host( lParts[ i ], lSizes[ i ], liKey, 0, -1 ) != MODGPU_OK ) return eError_InvalidData;
        return eError_NoError;
    }
    int liStatus = modgpu_cycle_parts_host( lParts.data(), lSizes.data(), (int)lParts.size(), liKey, liNumDevices );
    if( liStatus != MODGPU_OK )
    {
        std::printf( "ERROR: GPU part cipher failed: %s\n", modgpu_last_error() );
        return eError_InvalidData;
    }
    return eError_NoError;
}

// ------------------------------------------------------------------------------------ extract
eError CArk::ExtractFiles( int liFirstFileIndex, int liNumFiles, const char* lpTargetDirectory ) // CArk.cpp:424-504
{
    if( maFiles.empty() ) return eError_NoData;
    eError leError = LoadArkData();
    SHOW_ERROR_AND_RETURN;
    // The reference walks every entry whatever the two index arguments say (CArk.cpp:435; its only
    // caller passes (0, GetNumFiles())) and so does this by default; -fixquirks honours the range, clamped.
    int liBegin = 0, liEnd = (int)maFiles.size();
    if( CSettings::mbFixReferenceQuirks )
    {
        liBegin = std::max( 0, liFirstFileIndex );
        liEnd = (int)std::min< int64_t >( (int64_t)maFiles.size(), (int64_t)liBegin + std::max( 0, liNumFiles ) );
    }
    const std::string lTarget = lpTargetDirectory ? lpTargetDirectory : "";
    leError = ParallelOverEntries( liBegin, liEnd, [ & ]( int ii ) -> eError {
        const sFileDefinition& f = maFiles[ ii ];
        const std::string lOutputPath = lTarget + f.mName;
        if( KeepExisting( lOutputPath ) )
        {
            VERBOSE_OUT( "Output file already exists, skipping: " << lOutputPath << "\n" );
            return eError_NoError;
        }
        std::error_code ec;
        fs::path lParent = fs::path( lOutputPath ).parent_path();
        if( !lParent.empty() )
        {
            fs::create_directories( lParent, ec ); // concurrent creators of one directory are fine: checked below
            if( !fs::is_directory( lParent, ec ) ) return eError_FailedToCreateDirectory;
        }
        if( (uint64_t)f.mi64Offset + (uint64_t)f.miSize > maArkData.size() ) return eError_InvalidData;
        VERBOSE_OUT( "Writing file " << lOutputPath << "\n" );
        eError leWrite = WriteWholeFile( lOutputPath, maArkData.data() + f.mi64Offset, (size_t)f.miSize );
        if( leWrite == eError_FailedToCreateFile ) { std::printf( "Failed to create %s\n", lOutputPath.c_str() ); return eError_NoError; } // CArk.cpp:486-489
        return leWrite;
    } );
    SHOW_ERROR_AND_RETURN;
    return eError_NoError;
}

// ------------------------------------------------------------------------------------ build
bool CArk::ShouldPackFile( const std::vector< SSongConfig >& laSongs, const char* lpFilename ) const // CArk.cpp:57-92
{
    const char* lpSong = std::strstr( lpFilename, "/songs/" );
    if( !lpSong ) return true;
    const char* lpEnd = lpSong + 7;
    while( *lpEnd && *lpEnd != '/' ) ++lpEnd;
    if( !*lpEnd ) return false;
    std::string lSongDir( lpSong, lpEnd );
    std::transform( lSongDir.begin(), lSongDir.end(), lSongDir.begin(), []( unsigned char c ) { return (char)std::tolower( c ); } );
    for( const SSongConfig& s : laSongs )
        if( s.mPath.find( lSongDir ) != std::string::npos ) return true;
    return false;
}

static void AddBuiltInSongs( std::vector< SSongConfig >& laSongs ) // CArk.cpp:96-99, 762-765
{
    for( const char* lpPath : { "/songs/credits", "/songs/tut0", "/songs/tut1", "/songs/tutc" } )
    {
        SSongConfig s;
        s.mPath = lpPath;
        laSongs.push_back( s );
    }
}

static void ListFiles( const fs::path& lRoot, const std::string& lPrefix, std::vector< std::string >& lOut )
{
    // a directory's files first, then its sub-directories, each case-insensitively by name:
    // the order Utils.cpp:5-70 gets from FindFirstFileA on NTFS
    std::vector< std::pair< std::string, std::string > > lFiles, lDirs;
    std::error_code ec;
    for( const fs::directory_entry& e : fs::directory_iterator( lRoot, ec ) )
    {
        std::string lName = e.path().filename().string(), lKey = lName;
        std::transform( lKey.begin(), lKey.end(), lKey.begin(), []( unsigned char c ) { return (char)std::tolower( c ); } );
        ( e.is_directory( ec ) ? lDirs : lFiles ).emplace_back( lKey, lName );
    }
    std::sort( lFiles.begin(), lFiles.end() );
    std::sort( lDirs.begin(), lDirs.end() );
    for( auto& f : lFiles ) lOut.push_back( lPrefix + f.second );
    for( auto& d : lDirs ) ListFiles( lRoot / d.second, lPrefix + d.second + "/", lOut );
}

eError CArk::ConstructFromDirectory( const char* lpInputDirectory, const CArk& lReferenceHeader, std::vector< SSongConfig > laSongs ) // CArk.cpp:94-220
{
    AddBuiltInSongs( laSongs );
    const std::string lInput = WithSlash( lpInputDirectory );
    std::vector< std::string > laFilenames;
    ListFiles( lInput.empty() ? fs::path( "." ) : fs::path( lInput ), "", laFilenames );
    if( laFilenames.empty() ) { eError leError = eError_NoData; SHOW_ERROR_AND_RETURN; }
    VERBOSE_OUT( "Found " << laFilenames.size() << " files\n" );

    std::unordered_map< std::string, const sFileDefinition* > lKnown;
    for( const sFileDefinition& f : lReferenceHeader.maFiles ) lKnown.emplace( f.mName, &f ); // first of any duplicates

    maFiles.clear();
    uint64_t luTotalFileSize = 0;
    for( const std::string& lName : laFilenames )
    {
        auto lRef = lKnown.find( lName );
        if( lRef == lKnown.end() && CSettings::mbIgnoreNewFiles ) continue; // unknown files are dropped unless new files are allowed
        if( !CSettings::mbPackAllFiles && !ShouldPackFile( laSongs, lName.c_str() ) ) continue;
        std::error_code ec;
        auto luSize = fs::file_size( lInput + lName, ec );
        if( ec ) { std::printf( "Unable to open file: %s\n", lName.c_str() ); continue; }
        sFileDefinition lDef;
        if( lRef != lKnown.end() ) lDef = *lRef->second;
        lDef.mName = lName;
        lDef.miSize = (int)luSize;
        luTotalFileSize += luSize;
        maFiles.push_back( lDef );
    }
    if( lReferenceHeader.maArks.empty() ) { eError leError = eError_NoData; SHOW_ERROR_AND_RETURN; }
    // part names come from the reference header; sizes are an even plan of the total (CArk.cpp:207-217)
    maArks = lReferenceHeader.maArks;
    uint64_t luSizeRemaining = luTotalFileSize;
    for( size_t ii = 0; ii < maArks.size(); ++ii )
    {
        maArks[ ii ].muSize = (unsigned int)( luSizeRemaining / ( maArks.size() - ii ) );
        luSizeRemaining -= maArks[ ii ].muSize;
    }
    return eError_NoError;
}

eError CArk::ConstructFromTable( const std::vector< std::string >& laNames, const std::vector< unsigned int >& laSizes, int liNumArks, const char* lpArkPrefix )
{
    if( laNames.size() != laSizes.size() || liNumArks < 1 || liNumArks > kiMaxArks ) return eError_InvalidParameter;
    maFiles.clear();
    uint64_t luTotal = 0;
    for( size_t ii = 0; ii < laNames.size(); ++ii )
    {
        if( laNames[ ii ].empty() || laSizes[ ii ] > 0x7FFFFFFFu ) return eError_InvalidParameter;
        sFileDefinition lDef;
        lDef.mName = laNames[ ii ];
        lDef.miSize = (int)laSizes[ ii ];
        luTotal += laSizes[ ii ];
        maFiles.push_back( lDef );
    }
    maArks.assign( (size_t)liNumArks, sArkDefinition() );
    uint64_t luSizeRemaining = luTotal;
    for( int ii = 0; ii < liNumArks; ++ii )
    {
        maArks[ ii ].mPath = std::string( lpArkPrefix ? lpArkPrefix : "main" ) + "_" + std::to_string( ii ) + ".ark";
        maArks[ ii ].muSize = (unsigned int)( luSizeRemaining / (uint64_t)( liNumArks - ii ) );
        luSizeRemaining -= maArks[ ii ].muSize;
    }
    return eError_NoError;
}

// Offsets and part boundaries, from sizes alone (the bookkeeping half of CArk.cpp:783-823):
// files are laid back to back in table order; a part is closed after the file that takes it past
// its allowance, and the overshoot comes out of the next part's allowance.  64-bit here (F7).
eError CArk::SplitIntoArks()
{
    if( maArks.empty() ) return eError_NoData;
    size_t liArkIndex = 0;
    int64_t liAllowed = maArks[ 0 ].muSize;
    uint64_t luArkStart = 0, luPtr = 0;
    for( sFileDefinition& f : maFiles )
    {
        if( f.miSize == 0 ) { f.mi64Offset = 0; continue; }
        f.mi64Offset = (int64_t)luPtr;
        luPtr += (uint64_t)f.miSize;
        if( (int64_t)( luPtr - luArkStart ) > liAllowed && liArkIndex + 1 < maArks.size() )
        {
            uint64_t luArkSize = luPtr - luArkStart;
            if( luArkSize > 0xFFFFFFFFull ) return eError_ValueOutOfBounds; // header stores part sizes in 32 bits
            maArks[ liArkIndex ].muSize = (unsigned int)luArkSize;
            ++liArkIndex;
            liAllowed += (int64_t)maArks[ liArkIndex ].muSize - (int64_t)luArkSize;
            luArkStart = luPtr;
        }
    }
    if( luPtr - luArkStart > 0xFFFFFFFFull ) return eError_ValueOutOfBounds;
    maArks[ liArkIndex ].muSize = (unsigned int)( luPtr - luArkStart );
    for( size_t ii = liArkIndex + 1; ii < maArks.size(); ++ii ) maArks[ ii ].muSize = 0; // parts the plan never reached
    return eError_NoError;
}

eError CArk::BuildArk( const char* lpInputDirectory, std::vector< SSongConfig > laSongs ) // CArk.cpp:760-828
{
    AddBuiltInSongs( laSongs );
    VERBOSE_OUT( "Building ark\n" );
    eError leError = SplitIntoArks();
    SHOW_ERROR_AND_RETURN;
    uint64_t luTotalArkSize = 0;
    for( const sFileDefinition& f : maFiles ) luTotalArkSize += (uint64_t)f.miSize;
    if( !maArkData.Allocate( luTotalArkSize, PartSizes(), miPartDevices ) ) { leError = eError_NoData; SHOW_ERROR_AND_RETURN; } // CArk.cpp:780
    const std::string lInput = WithSlash( lpInputDirectory );
    leError = ParallelOverEntries( 0, (int)maFiles.size(), [ & ]( int ii ) -> eError {
        const sFileDefinition& f = maFiles[ ii ];
        if( f.miSize == 0 ) return eError_NoError;
        FILE* lpInputFile = std::fopen( ( lInput + f.mName ).c_str(), "rb" );
        if( !lpInputFile ) return eError_FailedToOpenFile;
        size_t got = std::fread( maArkData.data() + f.mi64Offset, 1, (size_t)f.miSize, lpInputFile );
        std::fclose( lpInputFile );
        return got == (size_t)f.miSize ? eError_NoError : eError_InvalidData;
    } );
    SHOW_ERROR_AND_RETURN;
    VERBOSE_OUT( "Ark built\n" );
    return eError_NoError;
}

eError CArk::BuildArkFromMemory( const char* lpData, uint64_t luDataSize )
{
    uint64_t luTotal = 0;
    for( const sFileDefinition& f : maFiles ) luTotal += (uint64_t)f.miSize;
    if( luTotal != luDataSize || ( luDataSize && !lpData ) ) return eError_InvalidParameter;
    eError leError = SplitIntoArks();
    ERROR_RETURN;
    if( !maArkData.Allocate( luDataSize, PartSizes(), miPartDevices ) ) return eError_NoData;
    // one thread copies 3.3 GB (BASELINE config 4) at ~10 GB/s; slices on several threads go at memory speed
    const uint64_t luSlice = 32ull << 20;
    const unsigned luThreads = (unsigned)std::min< uint64_t >( std::max( 1u, std::min( 16u, std::thread::hardware_concurrency() ) ), ( luDataSize + luSlice - 1 ) / luSlice );
    if( luThreads <= 1 )
    {
        if( luDataSize ) std::memcpy( maArkData.data(), lpData, (size_t)luDataSize );
        return eError_NoError;
    }
    std::atomic< uint64_t > lNext{ 0 };
    auto lCopy = [ & ]() {
        for( uint64_t luOff = lNext.fetch_add( luSlice ); luOff < luDataSize; luOff = lNext.fetch_add( luSlice ) )
            std::memcpy( maArkData.data() + luOff, lpData + luOff, (size_t)std::min< uint64_t >( luSlice, luDataSize - luOff ) );
    };
    std::vector< std::thread > lThreads;
    try { for( unsigned ii = 1; ii < luThreads; ++ii ) lThreads.emplace_back( lCopy ); } catch( ... ) {}
    lCopy();
    for( std::thread& t : lThreads ) t.join();
    return eError_NoError;
}

// ------------------------------------------------------------------------------------ save
eError CArk::SerialiseHeader( std::vector< unsigned char >& lOut, bool lbEncrypt ) const // CArk.cpp:901-1136
{
    const int liNumArks = (int)maArks.size();
    const int liNumFiles = (int)maFiles.size();
    lOut.clear();
    ByteSink out{ lOut };
    out.U32( CSettings::mbPS4 ? CSettings::kuEncryptedVersionPS4 : CSettings::kuEncryptedVersionPS3 ); // plaintext magic
    out.U32( kuUnencryptedVersion ); // sHeaderBase
    out.U32( 1 );
    out.Zero( 16 ); // mChecksumData: the reference leaves these 16 bytes uninitialised; zeros here
    out.U32( (uint32_t)liNumArks );
    out.U32( (uint32_t)liNumArks ); // ark sizes
    for( const sArkDefinition& a : maArks ) out.U32( a.muSize );
    out.U32( (uint32_t)liNumArks ); // ark paths
    for( const sArkDefinition& a : maArks ) out.Str( a.mPath );
    out.U32( (uint32_t)liNumArks ); // checksums, all zero
    out.Zero( 4 * (size_t)liNumArks );
    out.U32( (uint32_t)liNumArks ); // string counts, all zero
    out.Zero( 4 * (size_t)liNumArks );
    out.U32( (uint32_t)liNumFiles );

    // order of the entry table (CArk.cpp:969-1064)
    std::vector< int > lOrder( (size_t)liNumFiles ), lBucketOf( (size_t)liNumFiles );
    for( int ii = 0; ii < liNumFiles; ++ii )
    {
        lOrder[ ii ] = ii;
        lBucketOf[ ii ] = NameBucket( maFiles[ ii ].mName, liNumFiles );
    }
    if( CSettings::mbPS4 )
    {
        // PS4: by path -- at every level files before sub-directories, names case-insensitive, then
        // flags1, flags2.  The reference's comparator (CArk.cpp:977-1048) answers "true" for equal
        // keys, which std::sort does not allow; ties are broken by table position here instead.
        std::vector< std::vector< std::string > > lParts( (size_t)liNumFiles );
        for( int ii = 0; ii < liNumFiles; ++ii ) lParts[ ii ] = LowerComponents( maFiles[ ii ].mName );
        std::sort( lOrder.begin(), lOrder.end(), [ & ]( int a, int b ) {
            const auto &A = lParts[ a ], &B = lParts[ b ];
            for( size_t i = 0;; ++i )
            {
                const bool lbALeaf = i + 1 == A.size(), lbBLeaf = i + 1 == B.size();
                if( lbALeaf != lbBLeaf ) return lbALeaf;
                int c = A[ i ].compare( B[ i ] );
                if( c != 0 ) return c < 0;
                if( lbALeaf ) break;
            }
            if( maFiles[ a ].miFlags1 != maFiles[ b ].miFlags1 ) return maFiles[ a ].miFlags1 < maFiles[ b ].miFlags1;
            if( maFiles[ a ].miFlags2 != maFiles[ b ].miFlags2 ) return maFiles[ a ].miFlags2 < maFiles[ b ].miFlags2;
            return a < b;
        } );
    }
    else
    {
        // PS3: by bucket, then table position (CArk.cpp:1052-1063 compares the element addresses)
        std::sort( lOrder.begin(), lOrder.end(), [ & ]( int a, int b ) {
            return lBucketOf[ a ] != lBucketOf[ b ] ? lBucketOf[ a ] < lBucketOf[ b ] : a < b;
        } );
    }

    // chain links + entries (CArk.cpp:1066-1112): flags1 = table index of the previous entry in the
    // same bucket (-1 for the first), bucket head = the last such entry.  The reference finds them by
    // linear search over a vector; the result is the same map.
    std::unordered_map< int, int > lLast;
    const uint32_t luHashField = CSettings::mbPS4 ? 0xDDB682F0u : 0x7D401F60u; // CArk.cpp:719-720
    for( int liEntryIndex = 0; liEntryIndex < liNumFiles; ++liEntryIndex )
    {
        sFileDefinition& f = maFiles[ lOrder[ liEntryIndex ] ];
        const int liBucket = lBucketOf[ lOrder[ liEntryIndex ] ];
        auto lPrev = lLast.find( liBucket );
        f.miFlags1 = lPrev == lLast.end() ? -1 : lPrev->second;
        lLast[ liBucket ] = liEntryIndex;
        out.I64( f.mi64Offset ); // sFileDefinition::Serialise (CArk.cpp:685-721)
        out.Str( f.mName );
        out.U32( (uint32_t)f.miFlags1 );
        out.U32( (uint32_t)f.miSize );
        out.U32( f.miSize ? luHashField : 0u );
    }
    out.U32( (uint32_t)liNumFiles ); // bucket table (CArk.cpp:1114-1131)
    for( int ii = 0; ii < liNumFiles; ++ii )
    {
        auto lHead = lLast.find( ii );
        out.U32( (uint32_t)( lHead == lLast.end() ? -1 : lHead->second ) );
    }
    if( lOut.size() > 0xFFFFFFFFull ) return eError_ValueOutOfBounds;

    if( lbEncrypt )
    {
        CEncryptionCycler lEncrypt; // CArk.cpp:1135-1136: on save the key follows the platform switch
        lEncrypt.Cycle( lOut.data() + 4, (unsigned int)( lOut.size() - 4 ),
                        (int)( CSettings::mbPS4 ? CSettings::kuEncryptedPS4Key : CSettings::kuEncryptedPS3Key ) );
    }
    return eError_NoError;
}

eError CArk::SaveArk( const char* lpOutputDirectory, const char* lpHeaderFilename ) const // CArk.cpp:830-1192
{
    if( !lpHeaderFilename ) return eError_InvalidParameter;
    const std::string lOutput = lpOutputDirectory ? lpOutputDirectory : "";
    if( !CSettings::mbFixReferenceQuirks )
    {
        // the reference first insists that lpHeaderFilename can be opened for reading in the working
        // directory (CArk.cpp:904-909; it never reads from that handle).  "Working directory" is the process's
        // unless SetWorkingDirectory named one: a library must not chdir() a process other threads live in.
        FILE* lpHeaderFile = std::fopen( ( mWorkingDirectory + lpHeaderFilename ).c_str(), "rb" );
        if( !lpHeaderFile ) { eError leError = eError_FailedToOpenFile; SHOW_ERROR_AND_RETURN; }
        std::fclose( lpHeaderFile );
    }
    std::vector< unsigned char > lHeader;
    eError leError = SerialiseHeader( lHeader, true );
    SHOW_ERROR_AND_RETURN;
    const std::string lHeaderPath = lOutput + lpHeaderFilename;
    if( KeepExisting( lHeaderPath ) )
    {
        VERBOSE_OUT( "Output file already exists, skipping: " << lHeaderPath << "\n" );
    }
    VERBOSE_OUT( "Writing " << lHeaderPath << "\n" );
    leError = WriteWholeFile( lHeaderPath, lHeader.data(), lHeader.size() ); // written even when it exists (CArk.cpp:1160-1172)
    if( leError == eError_FailedToCreateFile )
    {
        VERBOSE_OUT( "Failed to open file for writing: " << lHeaderPath << "\n" );
        leError = eError_NoError;
    }
    SHOW_ERROR_AND_RETURN;

    // part slices (lSaveArk, CArk.cpp:845-899)
    struct sJob { std::string mFilename; uint64_t muSlice; unsigned int muSize; };
    std::vector< sJob > lJobs;
    uint64_t luPartStart = 0; // where a part's bytes are by the part sizes
    uint64_t luArkPtr = 0;    // the reference's lpArkPtr: moves only past parts it wrote (CArk.cpp:851, 891)
    for( const sArkDefinition& a : maArks )
    {
        const std::string lFilename = lOutput + a.mPath;
        // reference: a part that is skipped or cannot be opened leaves lpArkPtr where it was, so the
        // parts after it are written from the wrong slice (CArk.cpp:866, 883-891); -fixquirks uses the sizes
        const uint64_t luSlice = CSettings::mbFixReferenceQuirks ? luPartStart : luArkPtr;
        luPartStart += a.muSize;
        if( KeepExisting( lFilename ) )
        {
            std::printf( "Output file already exists: %s\n", lFilename.c_str() );
            continue;
        }
        if( luSlice + a.muSize > maArkData.size() ) { leError = eError_NoData; SHOW_ERROR_AND_RETURN; }
        std::error_code ec;
        std::printf( "%s %s\n", fs::exists( lFilename, ec ) ? "Overwriting" : "Writing", lFilename.c_str() ); // CArk.cpp:871-879
        fs::path lParent = fs::path( lFilename ).parent_path();
        if( !lParent.empty() ) fs::create_directories( lParent, ec );
        lJobs.push_back( { lFilename, luSlice, a.muSize } );
        luArkPtr += a.muSize;
    }
    if( !mbPartCipher ) // the reference's behaviour: raw slices
    {
        uint64_t luNotWritten = 0; // bytes of parts whose file could not be opened (reference mode: lpArkPtr stays behind)
        for( const sJob& j : lJobs )
        {
            leError = WriteWholeFile( j.mFilename, maArkData.data() + j.muSlice - luNotWritten, j.muSize );
            if( leError == eError_FailedToCreateFile )
            {
                std::printf( "Failed to open file for writing: %s\n", j.mFilename.c_str() );
                if( !CSettings::mbFixReferenceQuirks ) luNotWritten += j.muSize;
                leError = eError_NoError;
                continue;
            }
            SHOW_ERROR_AND_RETURN;
        }
        return eError_NoError;
    }
    // addition: each slice is streamed memory -> GPU -> file (H2D / kernel / D2H / pwrite overlapped,
    // modgpu_cycle_host_to_file); the in-memory buffer is left untouched.  Part i on GPU i mod N.
    const int liPartKey = (int)CSettings::Current().muKey; // on save the key follows the platform switch
    leError = ForEachPartOnDevices( lJobs.size(), miPartDevices, [ & ]( size_t ii, int liDevice ) {
        const sJob& j = lJobs[ ii ];
        if( liDevice < 0 ) // no GPU on this host: the slice goes through Cycle's host engine in pieces of a scratch buffer
        {                  // (the in-memory slice stays as it was), each piece at its own stream offset
            FILE* f = std::fopen( j.mFilename.c_str(), "wb" );
            if( !f ) return eError_FailedToCreateFile;
            constexpr uint64_t kuPiece = 64ull << 20;
            std::vector< uint8_t > lPiece( (size_t)std::min< uint64_t >( kuPiece, j.muSize ) );
            eError leResult = eError_NoError;
            for( uint64_t luAt = 0; luAt < j.muSize && leResult == eError_NoError; luAt += kuPiece )
            {
                const uint64_t luLen = std::min< uint64_t >( kuPiece, j.muSize - luAt );
                std::memcpy( lPiece.data(), maArkData.data() + j.muSlice + luAt, luLen );
                if( modgpu_cycle_auto_host( lPiece.data(), luLen, liPartKey, luAt, -1 ) != MODGPU_OK ) leResult = eError_InvalidData;
                else if( std::fwrite( lPiece.data(), 1, luLen, f ) != luLen ) leResult = eError_FailedToWriteData;
            }
            std::fclose( f );
            return leResult;
        }
        int liStatus = modgpu_cycle_host_to_file( reinterpret_cast< const uint8_t* >( maArkData.data() ) + j.muSlice, j.muSize,
                                                  j.mFilename.c_str(), liPartKey, 0, liDevice );
        if( liStatus == MODGPU_OK ) return eError_NoError;
        std::printf( "ERROR: part %s: %s\n", j.mFilename.c_str(), modgpu_last_error() );
        return liStatus == MODGPU_ERR_IO ? eError_FailedToWriteData : eError_InvalidData;
    } );
    SHOW_ERROR_AND_RETURN;
    return eError_NoError;
}
