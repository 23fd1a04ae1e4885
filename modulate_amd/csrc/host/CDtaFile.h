// CDtaFile.h -- binary DTA tree reader / writer, the slice of
// Modulate/CDtaFile.{h,cpp} that BASELINE config 1 touches: Load (CDtaFile.cpp:57-100, 393-509)
// and Save (CDtaFile.cpp:362-391, 1302-1326; leaf writers CDtaFile.h:262-284).  The reference's
// DTA files are PLAINTEXT (CDtaFile::Load never calls the cipher); config 1 wraps such a
// blob in the header framing to have something parseable behind the decrypt.  The song-list editing
// and .moggsong code of the reference are out of scope.
//
// Wire format (little-endian):
//   file   := u8 1, i32 1, tree-body, { i32 type(16|17), i32 1, tree-body }*
//   tree-body := i16 nChildren (> 0), i16 nodeId, child * nChildren
//   child  := i32 type, payload
//             0/6/8/9 -> i32        1 -> f32        5/18/33/35 -> i32 len, bytes
//             16/17   -> i32 1, tree-body
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "Error.h"

enum eNodeType // CDtaFile.h:10-23
{
    ENodeType_Integer0 = 0,
    ENodeType_Float = 1,
    ENodeType_String = 5,
    ENodeType_Integer6 = 6,
    ENodeType_Integer8 = 8,
    ENodeType_Integer9 = 9,
    ENodeType_Tree1 = 16,
    ENodeType_Tree2 = 17,
    ENodeType_Id = 18,
    ENodeType_IncludeFile = 33,
    ENodeType_Define = 35,
    ENodeType_Invalid
};

struct SDtaNode
{
    int miType = ENodeType_Tree1;
    short msNodeId = 0;             // trees only
    int miValue = 0;                // integer kinds
    float mfValue = 0.0f;           // ENodeType_Float
    std::string mString;            // string kinds
    std::vector< SDtaNode > maChildren; // trees only

    bool IsTree() const { return miType == ENodeType_Tree1 || miType == ENodeType_Tree2; }
};

class CDtaFile
{
public:
    eError Load( const char* lpFilename );
    eError Save( const char* lpFilename ) const;

    // additions: the same on memory images
    eError LoadFromMemory( const unsigned char* lpData, size_t liSize );
    void SaveToMemory( std::vector< unsigned char >& lOut ) const;

    const std::vector< SDtaNode >& GetTopLevel() const { return maTopLevel; }
    std::vector< SDtaNode >& GetTopLevel() { return maTopLevel; }
    std::string Dump() const; // one line per node, for tests

private:
    std::vector< SDtaNode > maTopLevel; // children of the reference's root node
};
