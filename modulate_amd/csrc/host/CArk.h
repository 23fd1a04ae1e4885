// CArk.h -- Linux host mirror of the reference's archive container for the cipher path.
//
// Public method names, argument meaning and eError behaviour follow Modulate/CArk.h:13-24,74;
// the storage behind them is std::vector / std::string (the reference's raw new[]/delete[]
// members, CArk.h:83-89, and its 25 000-file / 512 KiB-header limits, CArk.cpp:395-399, 911-912,
// are gone).  What this port covers is the buffer path SURVEY.md 8a lists:
//
//     Load            header file -> magic check -> Cycle(buf+4, size-4, key) -> parse   CArk.cpp:301-422
//     LoadArkData     all parts concatenated into one host buffer                        CArk.cpp:723-758
//     BuildArk        files -> one buffer, offsets, part split                           CArk.cpp:760-828
//     SaveArk         header serialise -> Cycle -> write; part slices written             CArk.cpp:830-1192
//     ExtractFiles    entries written out of the concatenated buffer                     CArk.cpp:424-504
//
// Every Cycle goes through CEncryptionCycler, i.e. the gfx950 kernel.  Additions that the
// reference does not have are grouped at the bottom and marked.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "Error.h"

// Modulate/CDtaFile.h:25-34 -- only the type is needed here (BuildArk / ConstructFromDirectory
// signatures); the song-list logic itself is out of scope.
struct SSongConfig
{
    std::string mId, mName, mUnlockMethod, mType, mPath, mArena;
    int miUnlockCount = -1;
};

// The concatenated part buffer (the reference's `char* mpArkData`, CArk.h:88, `new char[total]` at
// CArk.cpp:738, 780).  Allocated through modgpu_host_alloc / modgpu_host_alloc_parts: page-locked and device-visible when a GPU
// is present, so the part cipher DMAs straight from / to these pages (no staging copy); ordinary
// memory otherwise.  Contents are uninitialised after Allocate, as after the reference's new[].
class CArkDataBuffer
{
public:
    CArkDataBuffer() = default;
    ~CArkDataBuffer();
    CArkDataBuffer( const CArkDataBuffer& ) = delete;
    CArkDataBuffer& operator=( const CArkDataBuffer& ) = delete;
    // false: out of memory.  laPartSizes (optional; must add up to luSize): the parts that will be laid end to end in the
    // buffer -- their pages are then placed next to the GPU each part goes to (modgpu_host_alloc_parts).
    bool Allocate( uint64_t luSize, const std::vector< uint64_t >& laPartSizes = {}, int liNumDevices = 0 );
    void Release();
    char* data() { return mpData; }
    const char* data() const { return mpData; }
    uint64_t size() const { return muSize; }

private:
    char* mpData = nullptr;
    uint64_t muSize = 0;
};

class CArk
{
public:
    CArk();
    ~CArk();

    eError ConstructFromDirectory( const char* lpInputDirectory, const CArk& lReferenceHeader, std::vector< SSongConfig > laSongs );
    eError BuildArk( const char* lpInputDirectory, std::vector< SSongConfig > laSongs );
    eError SaveArk( const char* lpOutputDirectory, const char* lpHeaderFilename ) const;

    eError Load( const char* lpHeaderFilename );
    eError ExtractFiles( int liFirstFileIndex, int liNumFiles, const char* lpTargetDirectory );
    bool FileExists( const char* lpFilename ) const;

    int GetNumFiles() const;

    eError LoadArkData();

    // ------------------------------------------------------------------ additions (not in the reference)
    // Part-level cipher (BASELINE.json north_star; the reference stores parts raw):
    // every part slice of the concatenated buffer is cycled in place as its own stream from
    // offset 0, part i on GPU i mod liNumDevices (<= 0: all visible), no inter-GPU traffic.
    eError CycleArkData( int liKey, int liNumDevices = 0 );
    // When enabled, parts are ciphertext on disk: LoadArkData streams each part file disk -> GPU -> buffer
    // (key by the loaded header's magic) and SaveArk streams each slice buffer -> GPU -> file (key by
    // CSettings::mbPS4), part i on GPU i mod N, transfers and kernel overlapped.
    // Off by default = the reference's behaviour (parts stored raw).
    void EnablePartCipher( bool lbEnable, int liNumDevices = 0 ) { mbPartCipher = lbEnable; miPartDevices = liNumDevices; }

    // The directory the reference's bare-name opens resolve against (upstream: the process's working directory, where
    // main_<platform>.hdr lives -- Modulate.cpp:383-395, CArk.cpp:904-909).  Empty = the process's own.
    void SetWorkingDirectory( const std::string& lDirectory )
    {
        mWorkingDirectory = lDirectory;
        if( !mWorkingDirectory.empty() && mWorkingDirectory.back() != '/' ) mWorkingDirectory += '/';
    }

    // Synthetic table for BASELINE config 4: entries in the given order, liNumArks parts named
    // <lpArkPrefix>_<i>.ark with the reference's even size plan (CArk.cpp:207-217).
    eError ConstructFromTable( const std::vector< std::string >& laNames, const std::vector< unsigned int >& laSizes,
                               int liNumArks, const char* lpArkPrefix );
    // BuildArk with the file bytes supplied back-to-back in table order instead of read from disk.
    eError BuildArkFromMemory( const char* lpData, uint64_t luDataSize );

    // Header image exactly as SaveArk writes it (encrypted unless lbEncrypt is false).
    eError SerialiseHeader( std::vector< unsigned char >& lOut, bool lbEncrypt ) const;
    // Parse a header image (as read from disk: magic + encrypted body).
    eError ParseHeader( std::vector< unsigned char > lImage );

    int GetNumArks() const { return (int)maArks.size(); }
    unsigned int GetArkSize( int ii ) const { return maArks[ ii ].muSize; }
    const std::string& GetArkPath( int ii ) const { return maArks[ ii ].mPath; }
    const std::string& GetFileName( int ii ) const { return maFiles[ ii ].mName; }
    unsigned int GetFileSize( int ii ) const { return (unsigned int)maFiles[ ii ].miSize; }
    int64_t GetFileOffset( int ii ) const { return maFiles[ ii ].mi64Offset; }
    int GetFileFlags1( int ii ) const { return maFiles[ ii ].miFlags1; }
    int GetFileFlags2( int ii ) const { return maFiles[ ii ].miFlags2; }
    const char* GetArkData() const { return maArkData.data(); }
    bool IsArkDataPinned() const;
    uint64_t GetArkDataSize() const { return maArkData.size(); }

private:
    struct sArkDefinition // CArk.h:48-52
    {
        std::string mPath;
        unsigned int muSize = 0;
    };

    struct sFileDefinition // CArk.h:54-69 (wire fields; the std::hash members are replaced by name lookups)
    {
        std::string mName;
        int64_t mi64Offset = 0;
        int miFlags1 = -1;
        int miFlags2 = -1;
        int miHash = 0;
        int miSize = 0;
    };

    bool ShouldPackFile( const std::vector< SSongConfig >& laSongs, const char* lpFilename ) const;
    const sFileDefinition* GetFile( const std::string& lName ) const;
    eError SplitIntoArks();

    std::vector< sArkDefinition > maArks;
    mutable std::vector< sFileDefinition > maFiles; // SaveArk rewrites miFlags1, as the reference does (CArk.cpp:1094)
    CArkDataBuffer maArkData;
    bool mbPartCipher = false;
    int miPartDevices = 0;
    int miLoadedKey = 0;                            // key selected by the loaded header's magic
    std::vector< uint64_t > PartSizes() const; // maArks[].muSize
    std::string mHeaderDirectory; // where Load found the header: part paths resolve against it
    std::string mWorkingDirectory; // SetWorkingDirectory
    bool mbLoaded = false;
};
